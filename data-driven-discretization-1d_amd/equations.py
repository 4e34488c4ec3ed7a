"""Physics definitions: grids, random forcing and the 1-D PDE families.

Host-side mirror of ``pde_superresolution/equations.py``.  These objects are
*configuration*: they are evaluated once on the host to produce the constant
tables and scalar parameters the HIP kernels are launched with
(``Equation.kernel_spec()``), and they keep the reference's duck-typed NumPy
methods (``equation_of_motion``, ``finalize_time_derivative``, ``forcing``) so
that host code written against the reference keeps working.

Reference map:
  Grid                         equations.py:44-68
  Equation                     equations.py:71-193
  RandomForcing                equations.py:196-227
  Burgers / Conservative / Godunov     equations.py:230-370
  KdV / Conservative / Godunov         equations.py:373-478
  KS / Conservative / Godunov          equations.py:481-587
  type tables, from_hparams            equations.py:590-662
"""
import enum
import json
from typing import Mapping, Tuple

import numpy as np

from . import duckarray
from . import polynomials


@enum.unique
class ExactMethod(enum.Enum):
  """How the fine-grid "exact" solution of an equation is obtained."""
  POLYNOMIAL = 1
  SPECTRAL = 2
  WENO = 3


# Integer ids shared with the C ABI (include/ddd1d.h, enum ddd_equation).
KERNEL_EQ_BURGERS = 0
KERNEL_EQ_BURGERS_CONSERVATIVE = 1
KERNEL_EQ_KDV = 2
KERNEL_EQ_KDV_CONSERVATIVE = 3
KERNEL_EQ_KS = 4
KERNEL_EQ_KS_CONSERVATIVE = 5
KERNEL_EQ_BURGERS_GODUNOV = 6
KERNEL_EQ_KDV_GODUNOV = 7
KERNEL_EQ_KS_GODUNOV = 8


class Grid(object):
  """A periodic solution grid plus the finer reference grid it derives from."""

  def __init__(self, solution_num_points: int, resample_factor: int = 1,
               resample_method: str = 'mean', period: float = 1.0):
    self.resample_factor = resample_factor
    self.resample_method = resample_method
    self.period = period

    def _axis(n):
      dx = period / n
      return n, dx, dx * np.arange(n)

    (self.solution_num_points, self.solution_dx,
     self.solution_x) = _axis(solution_num_points)
    (self.reference_num_points, self.reference_dx,
     self.reference_x) = _axis(solution_num_points * resample_factor)

  def resample(self, x, axis: int = -1):
    """Reference resolution -> solution resolution."""
    return duckarray.RESAMPLE_FUNCS[self.resample_method](
        x, self.resample_factor, axis=axis)


class RandomForcing(object):
  """Sum of ``nparams`` travelling sine waves with seeded random parameters.

  The draw order (a, omega, k, phi) and the distributions follow
  equations.py:207-212 exactly so that a given seed yields the same forcing.
  """

  def __init__(self, grid: Grid, nparams: int = 20, seed: int = 0,
               amplitude: float = 1, k_min: int = 1, k_max: int = 3):
    self.grid = grid
    rs = np.random.RandomState(seed)
    shape = (nparams, 1)
    self.a = 0.5 * amplitude * rs.uniform(-1, 1, size=shape)
    self.omega = rs.uniform(-0.4, 0.4, size=shape)
    wavenumbers = np.arange(k_min, k_max + 1)
    self.k = rs.choice(np.concatenate([-wavenumbers, wavenumbers]), size=shape)
    self.phi = rs.uniform(0, 2 * np.pi, size=shape)

  def spatial_phase(self) -> np.ndarray:
    """float64 [nparams, reference_num_points]."""
    return 2 * np.pi * self.k * self.grid.reference_x / self.grid.period

  def __call__(self, t: float) -> np.ndarray:
    waves = np.sin(self.omega * t + self.spatial_phase() + self.phi)
    return self.grid.resample(np.sum(self.a * waves, axis=0))

  def export(self, path):
    """Text dump of the parameters (same row layout as the reference)."""
    meta = np.zeros_like(self.a)
    meta[0] = self.grid.period
    meta[1] = self.grid.reference_num_points
    np.savetxt(path, np.array(
        [self.a, self.omega, self.k, self.phi, meta]).squeeze())


def staggered_first_derivative(y, dx: float):
  """``(y[x+1] - y[x]) / dx`` with periodic wrap along the last axis."""
  y = np.asarray(y)
  shifted = np.concatenate([y[..., 1:], y[..., :1]], axis=-1)
  return (1 / dx) * (shifted - y)


def godunov_convective_flux(u_minus, u_plus):
  """Godunov flux for the convective term ``u**2 / 2``."""
  lo2, hi2 = u_minus ** 2, u_plus ** 2
  return 0.5 * np.where(u_minus <= u_plus,
                        np.minimum(lo2, hi2), np.maximum(lo2, hi2))


class Equation(object):
  """Base class: a periodic 1-D PDE ``u_t = F(u, u_x, ...)`` on a Grid.

  Class attributes every concrete equation defines:
    CONSERVATIVE, GRID_OFFSET, EXACT_METHOD, DERIVATIVE_NAMES,
    DERIVATIVE_ORDERS  -- as in the reference (equations.py:74-79)
    KERNEL_ID          -- which in-kernel equation of motion to use.
  """
  CONSERVATIVE = ...      # type: bool
  GRID_OFFSET = ...       # type: polynomials.GridOffset
  EXACT_METHOD = ...      # type: ExactMethod
  DERIVATIVE_NAMES = ...  # type: Tuple[str, ...]
  DERIVATIVE_ORDERS = ...  # type: Tuple[int, ...]
  KERNEL_ID = ...         # type: int

  def __init__(self, num_points: int, resample_factor: int = 1,
               period: float = 1.0, random_seed: int = 0):
    method = 'mean' if self.CONSERVATIVE else 'subsample'
    self.grid = Grid(num_points, resample_factor, method, period)
    self.random_seed = random_seed

  # -- interface -----------------------------------------------------------
  def initial_value(self) -> np.ndarray:
    raise NotImplementedError

  @property
  def time_step(self) -> float:
    """Step size for explicit fixed-step integration (midpoint rule)."""
    raise NotImplementedError

  @property
  def standard_deviation(self) -> float:
    """Empirical standard deviation of solutions (input normalisation)."""
    raise NotImplementedError

  def equation_of_motion(self, y, spatial_derivatives: Mapping[str, object]):
    raise NotImplementedError

  def finalize_time_derivative(self, t: float, y_t):
    """Hook applied only when integrating (not during training)."""
    del t
    return y_t

  def params(self) -> dict:
    raise NotImplementedError

  def to_fine(self) -> 'Equation':
    """Same equation and parameters on the reference-resolution grid."""
    return type(self)(**self.params())

  @classmethod
  def exact_type(cls):
    raise NotImplementedError

  @classmethod
  def conservative_type(cls):
    raise NotImplementedError

  @classmethod
  def base_type(cls):
    raise NotImplementedError

  def to_exact(self) -> 'Equation':
    # (an equation that already is its exact form: the object itself -- equations are
    # immutable, and rebuilding one re-seeds a RandomState, ~0.1 ms per sample that
    # integrate_exact_batch would pay for every seed of a batch)
    if type(self) is self.exact_type():
      return self
    return self.exact_type()(**self.params())

  def to_conservative(self) -> 'Equation':
    return self.conservative_type()(**self.params())

  # -- kernel-facing -------------------------------------------------------
  @property
  def has_time_dependent_forcing(self) -> bool:
    """True if finalize_time_derivative adds forcing(t)."""
    return False

  def kernel_spec(self) -> dict:
    """Scalars the HIP kernels need for this equation."""
    return dict(
        equation=self.KERNEL_ID,
        num_points=self.grid.solution_num_points,
        dx=self.grid.solution_dx,
        period=self.grid.period,
        eta=float(getattr(self, 'eta', 0.0)),
        standard_deviation=float(self.standard_deviation),
        conservative=bool(self.CONSERVATIVE),
        derivative_orders=tuple(self.DERIVATIVE_ORDERS),
        forced=self.has_time_dependent_forcing,
    )


class _RandomlyForcedEquation(Equation):
  """Shared plumbing for the three families (all own a RandomForcing)."""
  _DEFAULT_PERIOD = 1.0
  _FORCING_NPARAMS = 20
  _TIME_STEP = None
  _STANDARD_DEVIATION = None
  _EXTRA_PARAMS = ()   # names of additional constructor keywords

  def _setup(self, num_points, resample_factor, period, random_seed,
             k_min, k_max):
    Equation.__init__(self, num_points, resample_factor, period, random_seed)
    self.forcing = RandomForcing(self.grid, nparams=self._FORCING_NPARAMS,
                                 seed=random_seed, k_min=k_min, k_max=k_max)
    self.k_min = k_min
    self.k_max = k_max

  @property
  def time_step(self) -> float:
    return self._TIME_STEP

  @property
  def standard_deviation(self) -> float:
    return self._STANDARD_DEVIATION

  def params(self) -> dict:
    out = dict(num_points=self.grid.reference_num_points,
               period=self.grid.period,
               random_seed=self.random_seed)
    for name in self._EXTRA_PARAMS:
      out[name] = getattr(self, name)
    out.update(k_min=self.k_min, k_max=self.k_max)
    return out

  def _flux_divergence(self, flux):
    return -staggered_first_derivative(flux, self.grid.solution_dx)


# ---------------------------------------------------------------------------
# Burgers:  u_t + (u^2/2)_x = eta u_xx + f(x, t)
# ---------------------------------------------------------------------------
class BurgersEquation(_RandomlyForcedEquation):
  """Viscous Burgers' equation driven by random forcing; starts from rest."""
  CONSERVATIVE = False
  GRID_OFFSET = polynomials.GridOffset.CENTERED
  EXACT_METHOD = ExactMethod.WENO
  DERIVATIVE_NAMES = ('u_x', 'u_xx')
  DERIVATIVE_ORDERS = (1, 2)
  KERNEL_ID = KERNEL_EQ_BURGERS

  _FORCING_NPARAMS = 20
  _TIME_STEP = 1e-3
  _STANDARD_DEVIATION = 0.7917
  _EXTRA_PARAMS = ('eta',)

  def __init__(self, num_points: int, resample_factor: int = 1,
               period: float = 2 * np.pi, random_seed: int = 0,
               eta: float = 0.04, k_min: int = 1, k_max: int = 3):
    self._setup(num_points, resample_factor, period, random_seed, k_min, k_max)
    self.eta = eta

  def initial_value(self) -> np.ndarray:
    return np.zeros_like(self.grid.solution_x)

  def equation_of_motion(self, y, spatial_derivatives):
    u_x = spatial_derivatives['u_x']
    u_xx = spatial_derivatives['u_xx']
    return self.eta * u_xx - y * u_x

  def finalize_time_derivative(self, t, y_t):
    return y_t + self.forcing(t)

  @property
  def has_time_dependent_forcing(self) -> bool:
    return True

  @classmethod
  def exact_type(cls):
    return GodunovBurgersEquation

  @classmethod
  def conservative_type(cls):
    return ConservativeBurgersEquation

  @classmethod
  def base_type(cls):
    return BurgersEquation


class ConservativeBurgersEquation(BurgersEquation):
  """Burgers in flux form: u_t = -(u^2/2 - eta u_x)_x on staggered edges."""
  CONSERVATIVE = True
  GRID_OFFSET = polynomials.GridOffset.STAGGERED
  DERIVATIVE_NAMES = ('u', 'u_x')
  DERIVATIVE_ORDERS = (0, 1)
  KERNEL_ID = KERNEL_EQ_BURGERS_CONSERVATIVE

  def equation_of_motion(self, y, spatial_derivatives):
    del y
    u = spatial_derivatives['u']
    u_x = spatial_derivatives['u_x']
    return self._flux_divergence(0.5 * u ** 2 - self.eta * u_x)


class GodunovBurgersEquation(BurgersEquation):
  """Flux form with Godunov's upwind flux for the convective term."""
  CONSERVATIVE = True
  GRID_OFFSET = polynomials.GridOffset.STAGGERED
  DERIVATIVE_NAMES = ('u_minus', 'u_plus', 'u_x')
  DERIVATIVE_ORDERS = (0, 0, 1)
  KERNEL_ID = KERNEL_EQ_BURGERS_GODUNOV

  def equation_of_motion(self, y, spatial_derivatives):
    del y
    convective = godunov_convective_flux(spatial_derivatives['u_minus'],
                                         spatial_derivatives['u_plus'])
    return self._flux_divergence(
        convective - self.eta * spatial_derivatives['u_x'])


# ---------------------------------------------------------------------------
# Korteweg-de Vries:  u_t + 6 u u_x + u_xxx = 0
# ---------------------------------------------------------------------------
class KdVEquation(_RandomlyForcedEquation):
  """KdV with a random (sum of sines) initial condition, unforced."""
  CONSERVATIVE = False
  GRID_OFFSET = polynomials.GridOffset.CENTERED
  EXACT_METHOD = ExactMethod.SPECTRAL
  DERIVATIVE_NAMES = ('u_x', 'u_xxx')
  DERIVATIVE_ORDERS = (1, 3)
  KERNEL_ID = KERNEL_EQ_KDV

  _FORCING_NPARAMS = 10
  _TIME_STEP = 2.5e-5
  _STANDARD_DEVIATION = 0.594

  def __init__(self, num_points: int, resample_factor: int = 1,
               period: float = 32, random_seed: int = 0,
               k_min: int = 1, k_max: int = 3):
    self._setup(num_points, resample_factor, period, random_seed, k_min, k_max)

  def initial_value(self) -> np.ndarray:
    return self.forcing(0)

  def equation_of_motion(self, y, spatial_derivatives):
    u_x = spatial_derivatives['u_x']
    u_xxx = spatial_derivatives['u_xxx']
    return -6 * y * u_x - u_xxx

  @classmethod
  def exact_type(cls):
    return KdVEquation

  @classmethod
  def conservative_type(cls):
    return ConservativeKdVEquation

  @classmethod
  def base_type(cls):
    return KdVEquation


class ConservativeKdVEquation(KdVEquation):
  """KdV in flux form: u_t = -(3 u^2 + u_xx)_x."""
  CONSERVATIVE = True
  GRID_OFFSET = polynomials.GridOffset.STAGGERED
  DERIVATIVE_NAMES = ('u', 'u_xx')
  DERIVATIVE_ORDERS = (0, 2)
  KERNEL_ID = KERNEL_EQ_KDV_CONSERVATIVE

  def equation_of_motion(self, y, spatial_derivatives):
    del y
    u = spatial_derivatives['u']
    u_xx = spatial_derivatives['u_xx']
    return self._flux_divergence(3 * u ** 2 + u_xx)


class GodunovKdVEquation(KdVEquation):
  """Flux-form KdV with the Godunov convective flux."""
  CONSERVATIVE = True
  GRID_OFFSET = polynomials.GridOffset.STAGGERED
  DERIVATIVE_NAMES = ('u_minus', 'u_plus', 'u_xx')
  DERIVATIVE_ORDERS = (0, 0, 2)
  KERNEL_ID = KERNEL_EQ_KDV_GODUNOV

  def equation_of_motion(self, y, spatial_derivatives):
    del y
    convective = godunov_convective_flux(spatial_derivatives['u_minus'],
                                         spatial_derivatives['u_plus'])
    return self._flux_divergence(
        6 * convective + spatial_derivatives['u_xx'])


# ---------------------------------------------------------------------------
# Kuramoto-Sivashinsky:  u_t + u u_x + u_xx + u_xxxx = 0
# ---------------------------------------------------------------------------
class KSEquation(_RandomlyForcedEquation):
  """KS with a random (sum of sines) initial condition, unforced."""
  CONSERVATIVE = False
  GRID_OFFSET = polynomials.GridOffset.CENTERED
  EXACT_METHOD = ExactMethod.SPECTRAL
  DERIVATIVE_NAMES = ('u_x', 'u_xx', 'u_xxxx')
  DERIVATIVE_ORDERS = (1, 2, 4)
  KERNEL_ID = KERNEL_EQ_KS

  _FORCING_NPARAMS = 10
  _TIME_STEP = 2.5e-5
  _STANDARD_DEVIATION = 0.299

  def __init__(self, num_points: int, resample_factor: int = 1,
               period: float = 64, random_seed: int = 0,
               k_min: int = 1, k_max: int = 3):
    self._setup(num_points, resample_factor, period, random_seed, k_min, k_max)

  def initial_value(self) -> np.ndarray:
    return self.forcing(0)

  def equation_of_motion(self, y, spatial_derivatives):
    u_x = spatial_derivatives['u_x']
    u_xx = spatial_derivatives['u_xx']
    u_xxxx = spatial_derivatives['u_xxxx']
    return -y * u_x - u_xxxx - u_xx

  @classmethod
  def exact_type(cls):
    return KSEquation

  @classmethod
  def conservative_type(cls):
    return ConservativeKSEquation

  @classmethod
  def base_type(cls):
    return KSEquation


class ConservativeKSEquation(KSEquation):
  """KS in flux form: u_t = -(u^2/2 + u_xxx + u_x)_x."""
  CONSERVATIVE = True
  GRID_OFFSET = polynomials.GridOffset.STAGGERED
  DERIVATIVE_NAMES = ('u', 'u_x', 'u_xxx')
  DERIVATIVE_ORDERS = (0, 1, 3)
  KERNEL_ID = KERNEL_EQ_KS_CONSERVATIVE

  def equation_of_motion(self, y, spatial_derivatives):
    del y
    u = spatial_derivatives['u']
    u_x = spatial_derivatives['u_x']
    u_xxx = spatial_derivatives['u_xxx']
    return self._flux_divergence(0.5 * u ** 2 + u_xxx + u_x)


class GodunovKSEquation(KSEquation):
  """Flux-form KS with the Godunov convective flux."""
  CONSERVATIVE = True
  GRID_OFFSET = polynomials.GridOffset.STAGGERED
  DERIVATIVE_NAMES = ('u_minus', 'u_plus', 'u_x', 'u_xxx')
  DERIVATIVE_ORDERS = (0, 0, 1, 3)
  KERNEL_ID = KERNEL_EQ_KS_GODUNOV

  def equation_of_motion(self, y, spatial_derivatives):
    del y
    convective = godunov_convective_flux(spatial_derivatives['u_minus'],
                                         spatial_derivatives['u_plus'])
    return self._flux_divergence(
        spatial_derivatives['u_xxx'] + spatial_derivatives['u_x']
        + convective)


EQUATION_TYPES = {
    'burgers': BurgersEquation,
    'kdv': KdVEquation,
    'ks': KSEquation,
}

CONSERVATIVE_EQUATION_TYPES = {
    'burgers': ConservativeBurgersEquation,
    'kdv': ConservativeKdVEquation,
    'ks': ConservativeKSEquation,
}

FLUX_EQUATION_TYPES = {
    'burgers': GodunovBurgersEquation,
    'kdv': GodunovKdVEquation,
    'ks': GodunovKSEquation,
}


def equation_type_from_hparams(hparams):
  """(conservative, numerical_flux, equation) -> Equation subclass."""
  if not hparams.conservative:
    table = EQUATION_TYPES
  elif hparams.numerical_flux:
    table = FLUX_EQUATION_TYPES
  else:
    table = CONSERVATIVE_EQUATION_TYPES
  return table[hparams.equation]


def from_hparams(hparams, random_seed: int = 0) -> Tuple[Equation, Equation]:
  """Build the (fine, coarse) equation pair described by ``hparams``.

  ``equation_kwargs`` is a JSON object that must contain ``num_points`` (the
  fine grid size); the coarse grid has ``num_points // resample_factor``
  points and the division must be exact (equations.py:646-653).
  """
  kwargs = json.loads(hparams.equation_kwargs)
  fine_points = kwargs.pop('num_points')
  coarse_points, leftover = divmod(fine_points, hparams.resample_factor)
  if leftover:
    raise ValueError('resample_factor={} does not divide exact_num_points={}'
                     .format(hparams.resample_factor, fine_points))
  coarse = equation_type_from_hparams(hparams)(
      coarse_points, resample_factor=hparams.resample_factor,
      random_seed=random_seed, **kwargs)
  return coarse.to_fine(), coarse
