#!/usr/bin/env python3
"""One free-running BASELINE config-3 search (eps-greedy N = 64, K = 4, 18 sigma steps) in the f32 parity mode and in f16x3 from the same host
RNG: how many of the 72 selections coincide, where the first difference is and how decidable that decision was (bench.free_running_vs_f32).
~45 s on an MI355X.  Used to check a kernel change against the selections of the f32 mode (profiles/r05_experiments.txt items 4 and 7):
    python tools/free_running_probe.py          [DTS_CONV_EPI32=0|1 ...]"""
import sys, torch, json
sys.path.insert(0, '/root/repo')
import bench
class A: pass
a = bench.parse(['--no-cpu-baseline'])
job = bench.Job(a)
nets = {}
sd = None
for name in ('f32', 'f16x3'):
    n_, s_, sd = bench.build_adm(job, bench.torch_dtype(name), sd=sd)
    nets[name] = (n_, s_)
rec = bench.free_running_vs_f32(job, nets)
print(json.dumps(rec))
