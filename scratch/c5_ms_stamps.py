"""phases of pop_msym_kernel in one rank of C5 (a build with -DDC_MS_STAMPS after `git apply scratch/r6_msym_stamps.patch`; DC_LIB_PATH):
clock64 stamps summed over the waves, as fractions of the waves' total time"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, '.')
from clustering_amd import capi, density as dens
from clustering_amd.synth import gaussian_blobs
c = torch.from_numpy(gaussian_blobs(5_000_000, 30)).cuda()
radii = [0.30, 0.35, 0.40, 0.45, 0.50, 0.55, 0.60, 0.65]
buf = np.zeros(16, dtype=np.uint64)
dens.calculate_populations_segment(c, radii, 3, 8); torch.cuda.synchronize()
capi.lib.dc_dbg_ms_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(1))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(); dens.calculate_populations_segment(c, radii, 3, 8); ev[1].record(); torch.cuda.synchronize()
capi.lib.dc_dbg_ms_stamps(buf.ctypes.data_as(C.c_void_p), C.c_int(1))
names = ['vmcnt wait', 'barrier', 'fetch+flush check', 'reducer', 'tile head (ids, operands, chain 0, minima)', 'epilogue 0 + chain 1', 'finish 0', 'epilogue 1', 'finish 1', 'credit', 'scan']
tot = float(buf[11]); chains = float(buf[12])
print('ms', ev[0].elapsed_time(ev[1]), 'wave cycles per chain %.0f' % (tot / chains))
for n, v in zip(names, buf[:11]): print('%-48s %5.1f %%  %6.0f cycles per chain' % (n, 100.0 * float(v) / tot, float(v) / chains))
print('unaccounted %.1f %%' % (100.0 * (1.0 - float(buf[:11].sum()) / tot)))
