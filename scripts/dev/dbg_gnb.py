import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sklearn.naive_bayes import GaussianNB
from wdg_amd import ops
rng = np.random.default_rng(0)
n, f, c = 400, 300, 5
x = (rng.random((n, f)) ** 6).astype(np.float32); x[:, ::7] = 0; x /= x.sum(1, keepdims=True)
y = rng.integers(0, c, n).astype(np.int32)
tr, va = np.arange(0, 240, dtype=np.int32), np.arange(240, 400, dtype=np.int32)
gb = ops.GnbBatch([(torch.from_numpy(x).cuda(), torch.from_numpy(tr).cuda(), torch.from_numpy(va).cuda(), torch.from_numpy(y).cuda())], c, want_pred=True)
gb.launch(); torch.cuda.synchronize()
raw = gb.ws.cpu().numpy()
print("head", raw[:64].view(np.int32)[:10])
print("consts", raw[256:512].view(np.float64))
sk = GaussianNB().fit(x[tr], y[tr])
print("sk logprior", np.log(sk.class_prior_), "halflogdet", [-0.5 * np.sum(np.log(2 * np.pi * sk.var_[i])) for i in range(c)])
print("pred", gb.pred[0].cpu().numpy()[:20], "want", sk.predict(x[va])[:20])
print(sk._joint_log_likelihood(x[va])[:3])
