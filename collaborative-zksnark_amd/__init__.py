"""collaborative-zksnark_amd: MI355X (gfx950) engine for the per-party local compute of collaborative-zksnark's
MPC provers -- share-lane NTTs over BLS12-377 Fr and variable-base MSMs over BLS12-377 G1/G2 -- behind a C ABI
(include/czk.h).  This package is the thin Python face of that ABI used by tests/ and bench.py; the product is
libczk_hip.so (csrc/*.hip).  The directory name contains a hyphen, so it is imported through the repo-root
loader `czk_amd.py` (module name `czk_amd`).

There is no CPU fallback: importing `czk_amd.lib()` fails loudly if libczk_hip.so is missing, and nothing in
this package imports the checker under oracle/.
"""
from .binding import (CZK_FFT, CZK_IFFT, CZK_COSET_FFT, CZK_COSET_IFFT, CZK_MEM_HOST, CZK_MEM_DEVICE, CZK_MEM_NO_TABLES, CZK_MEM_ANY_POINTS, CZK_MEM_CHECK_SUBGROUP,  # noqa: F401
                      CZK_SCALAR_CANONICAL, CZK_SCALAR_MONTGOMERY, CZK_G1, CZK_G2, CzkError, Context, Bases, Net, CZK_NET_RCCL, CZK_NET_SHM, CZK_NET_IPC, lib,
                      lib_path, lab_lib, exported_symbols, header_symbols)
from . import binding  # noqa: F401
from .build import build  # noqa: F401
from . import parallel  # noqa: F401
