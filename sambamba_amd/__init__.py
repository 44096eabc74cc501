"""sambamba_amd -- Python harness around libsbx_depth.so, the MI355X (gfx950) engine behind
`sambamba depth base|region|window`.

The product is the C-ABI library (include/sbx_depth.h) plus the `sbx-depth` CLI, both built
from sambamba_amd/csrc/ by sambamba_amd.build.  This package only binds the C ABI with ctypes
for tests and bench.py; there is no Python or CPU implementation of the hot path, and loading
fails loudly when the library has not been built.
"""
from ._lib import (SbxError, Depth, lib, lib_path, inflate_blocks, compile_filter, regex_search, cli_path,  # noqa: F401
                   bgzf_compress, write_bam, build_index,
                   SBX_MODE_BASE, SBX_MODE_REGION, SBX_MODE_WINDOW)
