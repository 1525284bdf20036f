"""Probe (round 6): what a stage index PER WALKER SLOT with refill from the queue would buy the four-walkers-per-wave
local-energy kernel -- priced from measurements before building it.

Measured inputs: per walker of a settled production sweep (schedule order, first step from the learned table) the attempted steps and the
planned steps; the kernel's own wave-level evaluation count (validates the lockstep model); ticks per phase of the kernel
(profiles/r05_b_kbench_phase_stamps.json: 14 800 per wave-evaluation, consume 1 700, prologue + fused finish 1 180 amortised) and the consume
ticks BY STAGE of the -DFF_STAMPS_TRACE build (profiles/r06_a_kbench_consume_by_stage.json), scaled to that average.

Model of a walker: evaluation 1 = k0, then six evaluations (stages 1..6) per attempted step, one more (stage 0) after a rejected one; a walker
that attempted more steps than planned rejected its first.  Three machines on the same 2 048 resident waves fed from the same queue:
  L   lockstep (the kernel as it is): four walkers share a stage; a group ends with its slowest walker; one finish per group
  P1  a stage per slot, a slot refilled when its walker ends; the consume code runs once per DISTINCT stage present in the wave (divergent
      branches of one wave execute one after the other); the finish runs per slot (masked), i.e. once per walker
  P2  as P1 with the finish batched four at a time through a stash (state out to the workspace and back: + 10 % of a finish)
usage: python tools/probes/slot_refill_model.py [head|trained|driver1000|soak3000 ...]"""
import heapq, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as G
from fermiflow_amd import native

dev = torch.device("cuda:0")
B, WAVES = 65536, 2048
T_RHS = 14800.0 - 1700.0 - 1180.0
TRACE = {-2: 3734, 0: 834, 1: 968, 2: 1001, 3: 2323, 4: 1416, 5: 1343, 6: 4447}      # r06_a_kbench_consume_by_stage.json
SC = 1700.0 / ((TRACE[-2] + 2 * sum(TRACE[s] for s in range(1, 7))) / 13.0)
C = {s: v * SC for s, v in TRACE.items()}
FIN = 1180.0 * 14.0      # prologue + fused finish of one group of four (amortised 1 180 per evaluation over ~14 evaluations)


def stages_of(a, rej):
    """stage sequence of one walker: -2, then per attempt 1..6, a 0 after the rejected first attempt"""
    seq = [-2]
    for k in range(a):
        seq += [1, 2, 3, 4, 5, 6]
        if rej and k == 0:
            seq.append(0)
    return seq


def simulate(att, rej, mode):
    """event simulation: WAVES waves pull from one queue; returns (ticks until the last wave ends, wave-evaluations, mean distinct stages)"""
    n = len(att)
    nxt = 0
    heap = []
    state = {}
    for w in range(WAVES):
        slots = []
        for _ in range(4):
            if nxt < n:
                slots.append(stages_of(att[nxt], rej[nxt])[::-1]); nxt += 1
        if slots:
            state[w] = slots
            heapq.heappush(heap, (0.0, w))
    t_end, evals, distinct = 0.0, 0, 0
    while heap:
        t, w = heapq.heappop(heap)
        slots = state[w]
        if mode == "L":
            # lockstep: one evaluation per round position; the group runs max(len) evaluations (+ nothing else: a rejected walker's stage 0
            # is shared -- everyone passes through it)
            ln = max(len(s) for s in slots)
            anyrej = any(0 in s for s in slots)
            seq = stages_of((ln - 1 - (1 if anyrej else 0)) // 6, anyrej)
            dt = sum(T_RHS + C[s] for s in seq) + FIN
            evals += len(seq); distinct += len(seq)
            t += dt
            new = []
            for _ in range(4):
                if nxt < n:
                    new.append(stages_of(att[nxt], rej[nxt])[::-1]); nxt += 1
            if new:
                state[w] = new
                heapq.heappush(heap, (t, w))
            else:
                t_end = max(t_end, t)
            continue
        present = {s[-1] for s in slots if s}
        dt = T_RHS + sum(C[s] for s in present)
        evals += 1; distinct += len(present)
        done = 0
        for s in slots:
            if s:
                s.pop()
                if not s:
                    done += 1
        if done:
            # P1: the masked finish runs once for the slots that end in this evaluation; P2: a quarter of a batched finish per walker
            dt += FIN if mode == "P1" else 1.1 * FIN / 4.0 * done
            for k in range(len(slots)):
                if not slots[k] and nxt < n:
                    slots[k] = stages_of(att[nxt], rej[nxt])[::-1]; nxt += 1
        t += dt
        if any(slots):
            heapq.heappush(heap, (t, w))
        else:
            t_end = max(t_end, t)
    return t_end, evals, distinct / max(1, evals)


tags = sys.argv[1:] or ["head", "trained", "driver1000", "soak3000"]
W = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "trained_weights.npz"))
print(f"consume ticks by stage (scaled to the 1 700 average): " + " ".join(f"{s}:{v:.0f}" for s, v in C.items()) + f"; right-hand side {T_RHS:.0f}; finish per group {FIN:.0f}")
for tag in tags:
    model = G._model(dev, 3, 3, 2.0)
    if tag != "head":
        v = model.cnf.v_wrapper.v
        with torch.no_grad():
            for nm, m in (("eta", v.eta), ("mu", v.mu)):
                m.fc1.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w1"]).reshape(-1, 1))
                m.fc1.bias.copy_(torch.as_tensor(W[f"{tag}_{nm}_b1"]))
                m.fc2.weight.copy_(torch.as_tensor(W[f"{tag}_{nm}_w2"]).reshape(1, -1))
    cap = {}
    orig = native.eloc

    def eloc(*a, **k):
        cap["steps"] = torch.zeros(B, dtype=torch.int32, device=dev)
        cap["order"] = k.get("walker_order"); cap["hs"] = k.get("walker_h_init")
        k["walker_cost"] = cap["steps"]; k["want_stats"] = True
        r = orig(*a, **k)
        cap["stats"] = r["stats"]
        return r
    native.eloc = eloc
    torch.manual_seed(5)
    for it in range(12):
        with torch.no_grad():
            model(B)
    native.eloc = orig
    steps, order, cost = cap["steps"].cpu().numpy(), cap["order"].cpu().numpy(), model.walker_cost.cpu().numpy()
    hs = cap["hs"].cpu().numpy()
    light = (cost < 16)[order]
    att = steps[order][light].astype(int)
    planned = np.maximum(1, np.round(1.0 / np.maximum(hs[order][light], 1e-9))).astype(int)
    rej = att > planned
    solo = (1 + 6 * att + rej).mean()
    res = {m: simulate(att.tolist(), rej.tolist(), m) for m in ("L", "P1", "P2")}
    kern = cap["stats"][0].item() / B
    print(f"== {tag}: light walkers {len(att)}, attempted steps {att.mean():.2f}, first step rejected {rej.mean():.3f}; evaluations per walker alone {solo:.2f}; "
          f"kernel statistic (wave-level, incl. the routed heavy walkers) {kern:.2f}")
    for m, (t, ev, dist) in res.items():
        print(f"   {m}: wave-evaluations x 4 / walkers = {4 * ev / len(att):.2f}; distinct stages per evaluation {dist:.2f}; pass = {t / 2.4e6:.3f} ms at 2.4 GHz "
              f"({t / res['L'][0]:.3f} of lockstep)")
