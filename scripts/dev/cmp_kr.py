import numpy as np, sys
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
d = a.astype(int) - b.astype(int)
print("problems", len(a), "differing", int((d != 0).sum()), "max |diff|", int(np.abs(d).max()), "hist", np.bincount(np.abs(d))[:8])
