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
