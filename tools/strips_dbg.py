import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spfsplatv2_amd import synthetic as syn
from tests import util
for name, kw in (("bigtiles", dict(cfg="TESTBIG", S=1, V=1, G=6500)), ("k25", dict(cfg="TEST", S=2, V=2, G=1200, K=25, image_hw=(64, 48), s_mult=10.0))):
    res = []
    for strips in ("1", "2", "4"):
        os.environ["SPF_FWD_STRIPS"] = strips
        cfg = kw["cfg"]
        batch = syn.make_batch(cfg, kw["S"], kw["V"], seed=21, **{k: v for k, v in kw.items() if k not in ("cfg", "S", "V")})
        res.append(util.run_product(batch))
    a = res[0]
    for s_, b in zip(("2", "4"), res[1:]):
        same = all(torch.equal(a[k], b[k]) for k in ("color", "depth", "alpha")) and all(torch.equal(a["grads"][n], b["grads"][n]) for n in util.GRAD_NAMES)
        print(name, "strips", s_, "bit-identical:", same)
