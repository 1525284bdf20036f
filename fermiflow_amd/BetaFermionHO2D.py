"""Finite-temperature VMC driver with the reference's command line (src/BetaFermionHO2D.py:15-85)."""
import time

import torch

from . import HO2D, FreeFermion, MLP, Backflow, CNF, HO, CoulombPairPotential, BetaVMC


def main(argv=None):
    import argparse
    parser = argparse.ArgumentParser(description="Finite-temperature variational Monte Carlo simulation")
    parser.add_argument("--beta", type=float, default=10.0, help="inverse temperature")
    parser.add_argument("--nup", type=int, default=6, help="number of spin-up electrons")
    parser.add_argument("--ndown", type=int, default=0, help="number of spin-down electrons (must be 0)")
    parser.add_argument("--Z", type=float, default=0.5, help="Coulomb interaction strength")
    parser.add_argument("--deltaE", type=float, default=2.0, help="energy cutoff of the many-body states")
    parser.add_argument("--boltzmann", action="store_true", help="initialise the state weights to the Boltzmann distribution")
    parser.add_argument("--cuda", type=int, default=0, help="GPU device number")
    parser.add_argument("--Deta", type=int, default=50)
    parser.add_argument("--nomu", action="store_true")
    parser.add_argument("--Dmu", type=int, default=50)
    parser.add_argument("--t0", type=float, default=0.0)
    parser.add_argument("--t1", type=float, default=1.0)
    parser.add_argument("--iternum", type=int, default=1000)
    parser.add_argument("--batch", type=int, default=8000)
    args = parser.parse_args(argv)

    device = torch.device("cuda:%d" % args.cuda)
    torch.cuda.set_device(device)
    eta = MLP(1, args.Deta); eta.init_zeros()
    mu = None
    if not args.nomu:
        mu = MLP(1, args.Dmu); mu.init_zeros()
    cnf = CNF(Backflow(eta, mu=mu), (args.t0, args.t1))
    model = BetaVMC(args.beta, args.nup, args.ndown, args.deltaE, args.boltzmann, HO2D(), FreeFermion(device=device), cnf,
                    CoulombPairPotential(args.Z), sp_potential=HO())
    model.to(device=device)
    print("beta = %.1f, nup = %d, ndown = %d, Z = %.1f, Nstates = %d" % (args.beta, args.nup, args.ndown, args.Z, model.Nstates))
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-2)
    for i in range(1, args.iternum + 1):
        start = time.time()
        gradF_phi, gradF_theta = model(args.batch)
        optimizer.zero_grad()
        gradF_phi.backward()
        gradF_theta.backward()
        optimizer.step()
        torch.cuda.synchronize()
        print("iter: %03d" % i, "F:", model.F, "F_std:", model.F_std, "E:", model.E, "E_std:", model.E_std,
              "S:", model.S, "S_analytical:", model.S_analytical,
              "Instant speed (hours per 100 iters):", (time.time() - start) * 100 / 3600)


if __name__ == "__main__":
    main()
