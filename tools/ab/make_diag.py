#!/usr/bin/env python3
"""Builds tools/ab/libdts_diag.so: a DIAGNOSTIC copy of the conv kernel with s_memtime stamps (prologue / K loop / epilogue
sub-phases) written to the split-K workspace by thread 0 of every block.  Never shipped, never loaded by default."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, 'diffusion_tts_amd/csrc/conv_igemm.hip')).read()

def rep(old, new):
    global src
    assert old in src, old[:60]
    src = src.replace(old, new, 1)

rep("  const int nblk = p_n_ct * p_n_pt;\n  int bid = blockIdx.x;",
    "  const unsigned long long tsA = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();\n  const int nblk = p_n_ct * p_n_pt;\n  int bid = blockIdx.x;")
rep("  const int lrow = lane & 15, lq = lane >> 4;\n  int buf = 0;",
    "  const unsigned long long tsB = __builtin_amdgcn_s_memtime();\n  const int lrow = lane & 15, lq = lane >> 4;\n  int buf = 0;")
rep("                                                   int lq, char* smem, bool bias_in_acc) {",
    "                                                   int lq, char* smem, bool bias_in_acc, unsigned long long* estamp = nullptr) {")
rep("  V4 rv[MT][NT];\n  if constexpr (RES) {",
    "  if (estamp) estamp[2] = __builtin_amdgcn_s_memtime();\n  V4 rv[MT][NT];\n  if constexpr (RES) {")
rep("  constexpr bool want_stats = STATS && NT == 4;\n  float* sp = want_stats ? kp.stats + ((size_t)((pn0 + wn * 64) >> 6) * p_cout + cm0 + col0) * 2 : nullptr;",
    "  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  if (estamp) estamp[3] = __builtin_amdgcn_s_memtime();\n  constexpr bool want_stats = STATS && NT == 4;\n  float* sp = want_stats ? kp.stats + ((size_t)((pn0 + wn * 64) >> 6) * p_cout + cm0 + col0) * 2 : nullptr;")
rep("  __syncthreads();\n  // copy-out: 16 bytes per lane, whole rows;",
    "  if (estamp) estamp[0] = __builtin_amdgcn_s_memtime();\n  __syncthreads();\n  if (estamp) estamp[1] = __builtin_amdgcn_s_memtime();\n  // copy-out: 16 bytes per lane, whole rows;")
rep("                                              int lrow, int lq, char* smem, bool bias_in_acc) {",
    "                                              int lrow, int lq, char* smem, bool bias_in_acc, unsigned long long* estamp = nullptr) {")
rep("conv_epilogue_fast<T, MT, NT, BM, BN, R_, B_, S_>(kp, acc, cm0, pn0, wm, wn, lrow, lq, smem, bias_in_acc)",
    "conv_epilogue_fast<T, MT, NT, BM, BN, R_, B_, S_>(kp, acc, cm0, pn0, wm, wn, lrow, lq, smem, bias_in_acc, estamp)")
rep("  conv_epilogue<T, MT, NT, BM, BN>(kp, acc, cm0, pn0, (int)blockIdx.y, wm, wn, lrow, lq, smem, bias_in_acc);\n}", '''  const unsigned long long tsC = __builtin_amdgcn_s_memtime();
  unsigned long long est[4] = {0, 0, 0, 0};
  conv_epilogue<T, MT, NT, BM, BN>(kp, acc, cm0, pn0, (int)blockIdx.y, wm, wn, lrow, lq, smem, bias_in_acc, est);
  const unsigned long long tsD = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long tsE = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
  if (kp.splits == 1 && kp.partial != nullptr && threadIdx.x == 0) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(kp.partial) + (size_t)blockIdx.x * 12;
    d[0] = tsA; d[1] = tsB; d[2] = tsC; d[3] = est[0]; d[4] = est[1]; d[5] = tsD; d[6] = tsE; d[7] = rt0; d[8] = rt1; d[9] = est[2]; d[10] = est[3];
  }
}''')
tmp = '/tmp/conv_diag.hip'
open(tmp, 'w').write(src)
cs = os.path.join(ROOT, 'diffusion_tts_amd/csrc')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
                       '-I' + cs, '-c', tmp, '-o', '/tmp/conv_diag.o'], stderr=subprocess.DEVNULL)
objs = [os.path.join(cs, o) for o in ('conv_small.o', 'groupnorm.o', 'attention.o', 'elementwise.o')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(ROOT, 'tools/ab/libdts_diag.so'),
                       '/tmp/conv_diag.o', *objs])
print('built tools/ab/libdts_diag.so')
