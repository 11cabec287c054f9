"""Where the HOST time of a training step goes (cProfile over a few steps after warm-up).
    python scripts/host_profile.py joint 16"""
import cProfile, io, os, pstats, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]] + sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location("tsb", os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_step_bench.py"))
cfg, batch = sys.argv[1], sys.argv[2]
sys.argv = ["train_step_bench.py", cfg, "--batch", batch, "--steps", "1", "--warmup", "6"]
tsb = importlib.util.module_from_spec(spec)
# run main() once to build everything, but capture its `step` closure: re-implement minimal hook
src = open(spec.origin).read().replace("    for _ in range(a.warmup):", "    globals()['STEP'] = step\n    for _ in range(a.warmup):")
g = {"__name__": "tsb", "__file__": spec.origin}
exec(compile(src, spec.origin, "exec"), g)
g["main"]()
step = g["STEP"]
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("tottime")
st.print_stats(28)
st.print_callers("method 'to' of|synchronize|item|cpu'")
print(s.getvalue()[:6000])
