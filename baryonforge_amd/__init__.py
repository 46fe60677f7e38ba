"""
baryonforge_amd -- MI355X (gfx950) implementation of BaryonForge's per-halo shell
paint / baryonify hot path behind the reference's `bfg.Runners` / `bfg.Profiles` /
`bfg.utils` API for that path:

    import baryonforge_amd as bfg
    Cat   = bfg.utils.HaloLightConeCatalog(ra, dec, M, z, cosmo_dict)
    Shell = bfg.utils.LightconeShell(map=..., cosmo=cosmo_dict)
    model = bfg.utils.TabulatedProfile.from_arrays(ln1pz, lnM, lnr, table_2D)
    new   = bfg.Runners.PaintProfilesShell(Cat, Shell, epsilon_max=10, model=model).process()

Widened along SURVEY 8f: PaintProfilesAnisShell, BaryonifySnapshot (+ HaloNDCatalog,
ParticleSnapshot with NGP / CIC mass maps), the periodic-grid runners (+ GriddedMap)
and the device-side displacement-table builder.  The profile zoo, FFTLog pixel
windows and halo-model utilities are out of scope; see DESIGN.md.
"""
from . import Profiles, Runners, utils  # noqa: F401
from .Profiles import *  # noqa: F401,F403
from .Runners import *  # noqa: F401,F403
from .utils import *  # noqa: F401,F403
from .background import Background, MassDef  # noqa: F401

__version__ = "0.1.0"
