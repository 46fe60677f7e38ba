import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import baryonforge_amd as bfg
from baryonforge_amd import synthetic as syn
from baryonforge_amd.engine import get_context
from util import oracle_baryonify
warnings.simplefilter("ignore")
cosmo = dict(syn.COSMO)
ra, dec, M, z = syn.catalog(2000, seed=44)
zax, Max, rax, d = syn.displacement_table()
m_in = syn.mass_map(256); m_in[::7] = 0.0
Cat = bfg.HaloLightConeCatalog(ra, dec, M, z, cosmo)
model = bfg.Baryonification2D.from_arrays(zax, Max, rax, d, cosmo, epsilon_max=20)
off_ref, ptot = oracle_baryonify(cosmo, ra, dec, M, z, (zax, Max, rax), d, 256, 10, 20, None, offsets_only=True)
ctx = get_context()
# interleave a tile paint run to mimic the test order
zp = syn.pressure_table()
pm = bfg.TabulatedProfile.from_arrays(*zp)
for it in range(12):
    variant = ["scatter_wave", "scatter_quarter", "tile_lds"][it % 3]
    if it % 4 == 3:
        Sh = bfg.LightconeShell(map=np.zeros(12*1024*1024), cosmo=cosmo)
        bfg.PaintProfilesShell(Cat, Sh, 10, pm, verbose=False, variant="tile_lds").process()
    Shell = bfg.LightconeShell(map=m_in.copy(), cosmo=cosmo)
    R = bfg.BaryonifyShell(Cat, Shell, 10, model, verbose=False, variant=variant)
    d_off = R.offsets_device()
    off = d_off.cpu().numpy()
    err = np.abs(off - off_ref).max()
    d_in = ctx.to_device(m_in); d_out = ctx.zeros(m_in.size); d_sums = ctx.zeros(2)
    ctx.regrid_shell(256, d_off, d_in, d_out, d_sums)
    out = d_out.cpu().numpy(); sums = d_sums.cpu().numpy()
    print(it, variant, "off max err %.3e nan %d | sum in %.10e out %.10e  dev sums %s  stats %s" % (
        err, np.isnan(off).sum(), m_in.sum(), out.sum(), sums, R.last_stats), flush=True)
