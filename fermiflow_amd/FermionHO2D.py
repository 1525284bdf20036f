"""Ground-state VMC driver with the reference's command line (src/FermionHO2D.py:15-76).

    python -m fermiflow_amd.FermionHO2D --nup 3 --ndown 3 --Z 2.0 --batch 65536 --iternum 100
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m fermiflow_amd.FermionHO2D ...

Same flags, same objects, same loop; under torch.distributed (one process per GPU, RCCL) `--batch` is the global
walker count and every rank works on its shard (fermiflow_amd/dist.py).  `--save/--resume` add the
state_dict checkpoint the reference lacks; `--dim 3` the three-dimensional trap (HO3D orbitals: BASELINE.json configs[4] is
`--dim 3 --nup 10 --ndown 10 --sens_bits 32`), `--sens_bits 32` the single-precision sensitivity matrices of the matrix-core
local-energy kernel (11 particles and more; include/fermiflow.h, ff_set_sens_precision).
"""
import os
import time

import torch

from . import HO2D, HO3D, FreeFermion, MLP, Backflow, CNF, HO, CoulombPairPotential, GSVMC, checkpoint, native
from .utils import make_adam


def main(argv=None):
    import argparse
    parser = argparse.ArgumentParser(description="Ground-state variational Monte Carlo simulation")
    parser.add_argument("--nup", type=int, default=6, help="number of spin-up electrons")
    parser.add_argument("--ndown", type=int, default=0, help="number of spin-down electrons")
    parser.add_argument("--Z", type=float, default=0.5, help="Coulomb interaction strength")
    parser.add_argument("--cuda", type=int, default=0, help="GPU device number")
    parser.add_argument("--Deta", type=int, default=50, help="hidden layer size of the two-body backflow potential eta")
    parser.add_argument("--nomu", action="store_true", help="do not use the one-body backflow potential mu")
    parser.add_argument("--Dmu", type=int, default=50, help="hidden layer size of the one-body backflow potential mu")
    parser.add_argument("--t0", type=float, default=0.0, help="starting time")
    parser.add_argument("--t1", type=float, default=1.0, help="ending time")
    parser.add_argument("--iternum", type=int, default=1000, help="number of new iterations")
    parser.add_argument("--batch", type=int, default=8000, help="batch size (global, over all ranks)")
    parser.add_argument("--dim", type=int, default=2, choices=[2, 3], help="space dimension of the trap (not in the reference: 2 only)")
    parser.add_argument("--sens_bits", type=int, default=64, choices=[32, 64],
                        help="precision of the sensitivity matrices of the local-energy pass from 11 particles on (not in the reference)")
    parser.add_argument("--save", type=str, default=None, help="checkpoint file written after every iteration")
    parser.add_argument("--resume", type=str, default=None, help="checkpoint file to resume from")
    args = parser.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank = int(os.environ.get("RANK", "0"))
    device = torch.device("cuda:%d" % (local_rank if world > 1 else args.cuda))
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend="nccl", device_id=device)

    orbitals = HO2D() if args.dim == 2 else HO3D()
    native.set_sens_precision(args.sens_bits)
    basedist = FreeFermion(device=device)
    eta = MLP(1, args.Deta)
    eta.init_zeros()
    if not args.nomu:
        mu = MLP(1, args.Dmu)
        mu.init_zeros()
    else:
        mu = None
    v = Backflow(eta, mu=mu)
    cnf = CNF(v, (args.t0, args.t1))
    model = GSVMC(args.nup, args.ndown, orbitals, basedist, cnf, CoulombPairPotential(args.Z), sp_potential=HO())
    model.to(device=device)
    optimizer = make_adam(model.parameters(), lr=1e-2)
    start_iter = 1
    if args.resume:
        start_iter = checkpoint.load(args.resume, model, optimizer, device) + 1
    if rank == 0:
        print("nup = %d, ndown = %d, Z = %.1f" % (args.nup, args.ndown, args.Z))
        print("batch = %d, iternum = %d." % (args.batch, args.iternum))

    for i in range(start_iter, start_iter + args.iternum):
        start = time.time()
        gradE = model(args.batch)
        optimizer.zero_grad()
        gradE.backward()
        optimizer.step()
        torch.cuda.synchronize()
        speed = (time.time() - start) * 100 / 3600
        if rank == 0:
            print("iter: %03d" % i, "E:", model.E, "E_std:", model.E_std, "Instant speed (hours per 100 iters):", speed)
            if args.save:
                checkpoint.save(args.save, model, optimizer, i, device)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
