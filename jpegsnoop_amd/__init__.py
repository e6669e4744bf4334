"""jpegsnoop_amd -- MI355X-native JPEGsnoop scan-decode stage.

The product is `libjsnoop_gpu.so` (hand-written HIP kernels behind the C ABI of
include/jsnoop_gpu.h).  This package is its binding layer: `CimgDecode` mirrors the
reference's decoder object, `JpegBatch` is the batched submit.
"""
from .capi import load, last_error, LIB_PATH  # noqa: F401
from .imgdecode import CimgDecode, JpegBatch, JpegPipeline, dib_checksum_numpy  # noqa: F401
from .shard import partition_lpt, partition_contiguous, reduce_job_stats  # noqa: F401
