"""MI355X-native learned-stencil PDE integration (import as ``ddd1d_amd``).

Host-side mirror of the reference's integration path API
(``pde_superresolution.{equations,polynomials,layers,model,integrate}``) over
hand-written gfx950 HIP kernels behind a C ABI (include/ddd1d.h,
csrc/libddd1d.so).  See DESIGN.md.
"""
from . import duckarray
from . import polynomials
from . import equations
from . import hparams
from . import _lib
from . import layers
from . import model
from . import integrate
from . import distributed
from . import weno
from . import evaluation
from .hparams import (HParams, create_hparams, load_hparams, save_hparams,
                      checkpoint_dir_to_path)

__version__ = '0.1.0'
