"""src/zippy/common.nim:1-12 restated: error type, data formats, level names."""


class ZippyError(Exception):
    """Raised if an operation fails (common.nim:2).  `status` is the C-ABI
    status code (include/zippy_hip.h)."""

    def __init__(self, status, msg):
        super().__init__(msg)
        self.status = status


# CompressedDataFormat (common.nim:4-5), ordinals 0..3
dfDetect, dfZlib, dfGzip, dfDeflate = 0, 1, 2, 3

# common.nim:7-12
NoCompression = 0
BestSpeed = 1
BestCompression = 9
DefaultCompression = -1
HuffmanOnly = -2
