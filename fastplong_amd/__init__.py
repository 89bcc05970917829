"""fastplong_amd -- MI355X-native per-read hot path of long-read FASTQ preprocessing.

Package contents: `abi` (ctypes mirror of include/fastplong_amd.h), `engine` (the loader and
Python front end of the HIP library; fails loudly when the library or a GPU is missing),
`report` (summarize + fastplong.json writer working from the counter buffer), `synth`
(seeded synthetic read generators), `csrc/` (HIP kernels + C-ABI), `host/` (C++ host: FASTQ
reader/writer, batcher, CLI).
"""
__version__ = "0.1.0"
