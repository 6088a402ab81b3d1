"""extendablesparse.jl_amd -- MI355X-native sparse-assembly backend for ExtendableSparseMatrix.

Only what the hot path needs: csrc/ (HIP kernels + the C ABI of include/esparse_hip.h) and the
host-side mirror of the reference's operator interface.  Importing the package never touches the
oracle and never falls back to a CPU implementation.
"""
from . import _lib
from ._lib import (ESP_COO, ESP_FLUSH_PLUS, ESP_FLUSH_ROUTED, ESP_OP_ADD, ESP_OP_SUB, ESP_RAWUPDATE, ESP_SET,
                   ESP_UPDATE, BoundsError, EspError, NoDeviceError)
from .matrix import (ExtendableSparseMatrix, GenericExtendableSparseMatrixCSC,
                     GenericMTExtendableSparseMatrixCSC, SparseMatrixCSC, SparseMatrixHIPCOO)
from . import fdrand as fdrand_module
from .fdrand import fdrand, fdrand_, fdrand_coo, fdrand_device_
from .sharded import GroupShardedMatrix, owner_ranges

# aliases mirroring src/ExtendableSparse.jl:34-39
ExtendableSparseMatrixCSC = ExtendableSparseMatrix
HIPExtendableSparseMatrixCSC = GenericExtendableSparseMatrixCSC
MTHIPExtendableSparseMatrixCSC = GenericMTExtendableSparseMatrixCSC


def library_path():
    return _lib.SO
