"""Byte view of raw device memory as a torch tensor (tests only)."""


class _Dev:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def view_u8(torch, ptr, nbytes):
    return torch.as_tensor(_Dev(ptr, int(nbytes)), device="cuda")
