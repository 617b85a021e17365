"""insert clock64 phase stamps into the working tree's pop_msym_kernel (measurement build, -DDC_MS_STAMPS; `git checkout` the two
files afterwards): scratch/c5_ms_stamps.py reads them"""
import sys
p = 'clustering_amd/csrc/dc_mfma_msym.hpp'
s = open(p).read()
def rep(old, new, cnt=1):
    global s
    assert s.count(old) >= 1, old[:60]
    s = s.replace(old, new)
rep("template <int NM, int NR, bool INPL>\n__global__ __launch_bounds__(256, 2) void pop_msym_kernel(", '''#ifdef DC_MS_STAMPS
__device__ unsigned long long g_ms_dbg[16];
#define MS_STAMP(c) do { const unsigned long long now_ = clock64(); dbg_acc[c] += now_ - dbg_last; dbg_last = now_; } while (0)
#else
#define MS_STAMP(c) do { } while (0)
#endif
template <int NM, int NR, bool INPL>
__global__ __launch_bounds__(256, 2) void pop_msym_kernel(''')
rep("  const PopSetup<NR> P = pop_setup<NR>(hdr, rad2, n_cols);\n  float r2max = rad2.v[0];", '''#ifdef DC_MS_STAMPS
  unsigned long long dbg_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long dbg_t0 = clock64();
  unsigned long long dbg_last = dbg_t0;
#endif
  const PopSetup<NR> P = pop_setup<NR>(hdr, rad2, n_cols);
  float r2max = rad2.v[0];''')
rep("    if (lane == 0) list_cnt[wib] = cnt;\n    __syncthreads();", "    if (lane == 0) list_cnt[wib] = cnt;\n    __syncthreads();\n    MS_STAMP(10);")
for line in s.split('\n'):
    pass
import re
# window top
s = re.sub(r"(\n *)__builtin_amdgcn_s_waitcnt\(0x0F70\);   // vmcnt\(0\): this wave's share of the window starting at i0?\n( *)__syncthreads\(\);", lambda m: m.group(1) + "MS_STAMP(9);" + m.group(1) + "__builtin_amdgcn_s_waitcnt(0x0F70);" + m.group(1) + "MS_STAMP(0);\n" + m.group(2) + "__syncthreads();\n" + m.group(2) + "MS_STAMP(1);", s, count=1)
rep("#ifndef DC_MS_ABL_NOREDUCE\n", "          MS_STAMP(2);\n#ifndef DC_MS_ABL_NOREDUCE\n")
s = s.replace("std::true_type{});\n          }\n#endif\n", "std::true_type{});\n          }\n#endif\n          MS_STAMP(3);\n", 1)
# the in-place instance: head (operands, both Gram chains, minima), the two epilogues in turn, finish 0, finish 1
rep("            const int k0 = skip_count(acc0), k1 = skip_count(acc1);\n", "            const int k0 = skip_count(acc0), k1 = skip_count(acc1);\n            MS_STAMP(4);\n")
rep("            t = (uint32_t)__builtin_amdgcn_readfirstlane(t_raw);\n            finish(std::integral_constant<int, 0>{}, e, t);\n            finish(std::integral_constant<int, 1>{}, e1, t);", "            MS_STAMP(5);\n            t = (uint32_t)__builtin_amdgcn_readfirstlane(t_raw);\n            finish(std::integral_constant<int, 0>{}, e, t);\n            MS_STAMP(6);\n            finish(std::integral_constant<int, 1>{}, e1, t);\n            MS_STAMP(8);")
s = re.sub(r"(\n *credit\([^\n]*\);\n)", lambda m: m.group(1) + "          MS_STAMP(9);\n", s, count=1)
rep("  if (lane == 0 && chain_counter && wave_live) {\n    atomicAdd(chain_counter, (unsigned long long)chains);", '''#ifdef DC_MS_STAMPS
  if (lane == 0 && wave_live) {
    for (int c_ = 0; c_ < 11; ++c_) atomicAdd(&g_ms_dbg[c_], dbg_acc[c_]);
    atomicAdd(&g_ms_dbg[11], clock64() - dbg_t0);
    atomicAdd(&g_ms_dbg[12], (unsigned long long)chains);
  }
#endif
  if (lane == 0 && chain_counter && wave_live) {
    atomicAdd(chain_counter, (unsigned long long)chains);''')
open(p, 'w').write(s)
p = 'clustering_amd/csrc/dc_mfma_step.hip'
s = open(p).read()
s += '''
#if defined(DC_MS_STAMPS) && DC_STEP == 6
extern "C" __attribute__((visibility("default"))) int dc_dbg_ms_stamps(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dc::g_ms_dbg), sizeof(unsigned long long) * 16, 0, hipMemcpyDeviceToHost);
  if (reset) { unsigned long long z[16] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(dc::g_ms_dbg), z, sizeof(z), 0, hipMemcpyHostToDevice); }
  return rc;
}
#endif
'''
open(p, 'w').write(s)
print('stamps:', open('clustering_amd/csrc/dc_mfma_msym.hpp').read().count('MS_STAMP('))
