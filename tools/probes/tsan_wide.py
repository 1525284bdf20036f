"""One launch of the wide local-energy kernel in the host simulator, for the sanitizer builds of tests/hostsim (make -C tests/hostsim tsan):
  tsan_wide.py NUP NDN DIM BITS"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.hostsim import simlib as S
nup, ndn, d, bits = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
n = nup + ndn
rng = np.random.default_rng(7 + n)
He = Hm = 8
eta = [rng.normal(size=He) * 0.5, rng.normal(size=He) * 0.3, rng.normal(size=He) * 0.05]
mu = [rng.normal(size=Hm) * 0.5, rng.normal(size=Hm) * 0.3, rng.normal(size=Hm) * 0.05]
x = rng.normal(size=(2, n, d)) * 0.7
net = S.Net(eta, mu, table=True)
S.lib().ff_set_sens_precision(bits)
r = (S.eloc3d if d == 3 else S.eloc)(x, nup, ndn, net, 2.0, rtol=1e-4, atol=1e-6)
print("eloc", r["eloc"], r["stats"][:4])
