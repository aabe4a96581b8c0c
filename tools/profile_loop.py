import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
exec(open(os.path.join(ROOT, "tools", "_loop_scene.py")).read())
cfg2 = dict(config); cfg2["max_iterations"] = 10
f = FusedRenderAndCompare(dec, cam, cfg2, targets)
for rep in range(3):
    f(*args[1:], use_graph=True)
torch.cuda.synchronize()
