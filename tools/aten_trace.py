import sys, argparse, traceback, collections
sys.path[:0] = ["sub-gc_amd", "."]
import torch, bench
import subgc.models as models
from subgc import synthetic, parallel
from torch.utils._python_dispatch import TorchDispatchMode
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "full_gc_kar"]
dev = torch.device("cuda:0")
m = models.setup(argparse.Namespace(**cfg["opt"])).to(dev).train()
lw = models.LossWrapper(m, None)
b = {k: v.to(dev) for k, v in synthetic.make_train_batch(8, seed=1000, **cfg["data"]).items()}
adam = parallel.FlatAdam(m)
one = torch.ones((), device=dev)
def step():
    m.flatten_grads(); out = lw(*bench.lw_args(b)); models.total_loss(out).backward(one); adam.step()
step(); step()
seen = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        r = func(*args, **(kwargs or {}))
        name = str(func)
        launches = any(s in name for s in ("fill", "zero", "copy_", "add", "mul", "sum", "cat", "index", "clone", "_to_copy", "ones", "full", "arange", "sort", "gather", "scatter", "where", "eq", "lt", "gt", "div", "sub"))
        if launches:
            t = r if torch.is_tensor(r) else (args[0] if args and torch.is_tensor(args[0]) else None)
            if t is not None and t.is_cuda:
                fr = [f for f in traceback.extract_stack() if "/subgc/" in f.filename or "bench.py" in f.filename]
                where = f"{fr[-1].filename.split('/')[-1]}:{fr[-1].lineno}" if fr else "?"
                seen[(name, str(t.dtype), where)] += 1
        return r
with Log():
    step()
torch.cuda.synchronize()
for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(v, k)
