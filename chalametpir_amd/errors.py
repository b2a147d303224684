"""ChalametPIRError: the reference's error enum (chalametpir_common/src/error.rs:7-48) on the Python side.

`code` is the cpir_status returned by the C ABI (include/chalamet_hip.h); `variant` is the name of the reference enum
variant it stands for, so parity tests can assert on the same names the reference's tests match on."""
from __future__ import annotations

VARIANTS = {
    1: "InvalidMatrixDimension",
    2: "IncompatibleDimensionForMatrixMultiplication",
    4: "InvalidNumberOfElementsInMatrix",
    5: "IncompatibleDimensionForRowVectorTransposedMatrixMultiplication",
    7: "FailedToDeserializeMatrixFromBytes",
    8: "EmptyKVDatabase",
    9: "ExhaustedAllAttemptsToBuild3WiseXorFilter",
    10: "ExhaustedAllAttemptsToBuild4WiseXorFilter",
    14: "KVDatabaseSizeTooLarge",
    17: "UnsupportedArityForBinaryFuseFilter",
    19: "ImpossibleEncodedDBMatrixElementBitLength",
    # no reference analogue beyond the Vulkan* family (error.rs:10-22) they replace
    64: "HipNoDevice",
    65: "HipRuntimeCallFailed",
    66: "HipOutOfMemory",
    67: "BufferTooSmall",
    68: "InvalidArgument",
    69: "ShardRange",
}


class ChalametPIRError(Exception):
    def __init__(self, code: int, message: str = "", detail: str = ""):
        self.code = int(code)
        self.variant = VARIANTS.get(self.code, f"Unknown({code})")
        text = f"{self.variant}: {message}" if message else self.variant
        if detail:
            text += f" [{detail}]"
        super().__init__(text)

    def __eq__(self, other):  # the reference derives PartialEq (error.rs:7)
        return isinstance(other, ChalametPIRError) and other.code == self.code

    def __hash__(self):
        return hash(self.code)
