#!/usr/bin/env python3
"""Static check of compiler-generated AGPR copies in gfx950 assembly (hipcc -S --cuda-device-only).

The local-energy kernels live at 256 VGPRs + up to 232 AGPRs, so the register allocator parks values in AGPRs
(v_accvgpr_write / v_accvgpr_read).  ROCm 7.2's LLVM was caught placing such a copy INSIDE a conditionally executed
block (the `if (valid && A.h_init)` warm-start block of ff_ode_fwd_kernel<2,2,2,true>) while the matching read sits
behind the join: when no lane takes the branch (s_cbranch_execz) the AGPR is never written and the read returns
garbage -- an LDS address in that case, HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION at run time.

More precisely the copy sits at the top of the JOIN block, in front of the `s_or_b64 exec, exec, <saved>` that
re-enables the lanes which skipped the branch: it executes under the narrowed mask (exec = 0 when nobody took the branch).

Check 1 (default, exit status 1 on a hit): every AGPR written between a block label and the block's
`s_or_b64 exec, exec, s[..]` whose value is read anywhere else in the kernel.
Check 2 (--paths): control-flow graph from labels and s_branch / s_cbranch_*, "definitely written" dataflow over the
AGPRs, every AGPR read that some path from the kernel entry reaches without a write.  Path-level only, and noisy: the
radial-table rows are loaded straight into AGPRs under the `r < r_max` guard and copied out behind it (dead there).

usage: check_agpr_spills.py [--paths] file.s|lib.so [kernel-name-substring]
       (a .so is taken apart into its gfx950 code objects and disassembled with llvm-objdump: seconds for the whole library;
        tests/test_host_logic.py runs exactly that on the built libfermiflow_hip.so)
"""
import os, re, subprocess, sys, tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
READ_ONLY_FIRST = ("ds_write", "ds_add", "ds_max", "ds_min", "global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "ds_cmpst")


def aregs(tok):
    out = set()
    for m in AREG.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(path):
    funcs, cur, name = {}, None, None
    for raw in open(path):
        line = raw.split(";")[0].split("//")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(\S+):\s*$", line)
        if m and not line.startswith("\t") and not line.startswith(" "):
            lab = m.group(1)
            if not lab.startswith(".L") and (lab.startswith("_Z") or "kernel" in lab):
                name, cur = lab, []
                funcs[name] = cur
                continue
            if cur is not None:
                cur.append(("label", lab))
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith("."):
            if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") or s.startswith(".Lfunc_end"):
                cur = None if s.startswith(".section") else cur
            continue
        cur.append(("ins", s))
    return funcs


def parse_library(path):
    """{kernel: items} of every gfx950 code object embedded in a host shared library (.hip_fatbin section)."""
    data = open(path, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)][1:]      # [0] is the host ELF itself
    funcs = {}
    with tempfile.TemporaryDirectory() as tmp:
        for k, st in enumerate(starts):
            end = starts[k + 1] if k + 1 < len(starts) else len(data)
            co = os.path.join(tmp, "co%d.elf" % k)
            open(co, "wb").write(data[st:end])
            r = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", co], capture_output=True, text=True)
            if r.returncode != 0 or "elf64-amdgpu" not in r.stdout:
                continue
            funcs.update(parse_objdump(r.stdout))
    return funcs


def parse_objdump(text):
    funcs, name, base, ins = {}, None, 0, []
    def finish():
        if name is None: return
        targets = set()
        for addr, s_, tgt in ins:
            if tgt is not None: targets.add(tgt)
        items = []
        for addr, s_, tgt in ins:
            if addr in targets: items.append(("label", "L%x" % addr))
            op = s_.split()[0]
            if tgt is not None:      # branch: replace the numeric operand by the label
                s_ = " ".join(s_.split()[:-1] + ["L%x" % tgt]) if len(s_.split()) > 1 else s_
            items.append(("ins", s_))
        funcs[name] = items
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            finish()
            base, name, ins = int(m.group(1), 16), m.group(2), []
            continue
        if name is None or "//" not in line: continue
        code, comment = line.split("//", 1)
        code = code.strip()
        if not code: continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not am: continue
        addr = int(am.group(1), 16)
        tgt = None
        if code.startswith(("s_cbranch", "s_branch")):
            tm = re.search(r"<[^>+]+\+0x([0-9a-fA-F]+)>", comment)
            tgt = base + int(tm.group(1), 16) if tm else (base if re.search(r"<[^>+]+>", comment) else None)
        ins.append((addr, code, tgt))
    finish()
    return funcs


def check(name, items):
    # basic blocks
    blocks, labels, cur = [], {}, []
    def flush():
        nonlocal cur
        if cur:
            blocks.append(cur); cur = []
    for kind, s in items:
        if kind == "label":
            flush(); labels[s] = len(blocks)
            continue
        cur.append(s)
        op = s.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64"):
            flush()
    flush()
    n = len(blocks)
    succ = [[] for _ in range(n)]
    for i, b in enumerate(blocks):
        last = b[-1].split()
        op = last[0]
        if op == "s_endpgm":
            continue
        if op == "s_branch":
            t = labels.get(last[1]);  succ[i] += [t] if t is not None else []
            continue
        if op.startswith("s_cbranch"):
            t = labels.get(last[-1])
            if t is not None: succ[i].append(t)
        if i + 1 < n: succ[i].append(i + 1)
    pred = [[] for _ in range(n)]
    for i in range(n):
        for t in succ[i]: pred[t].append(i)
    # per-instruction defs/uses
    def du(s):
        parts = s.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if not ops: return set(), set()
        if op.startswith(READ_ONLY_FIRST):
            return set(), set().union(*[aregs(o) for o in ops])
        d = aregs(ops[0])
        u = set().union(*[aregs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        if "mfma" in op and len(ops) >= 4: u |= aregs(ops[3])
        return d, u
    ALL = set(range(512))
    IN = [set(ALL) for _ in range(n)]; OUT = [set(ALL) for _ in range(n)]
    if n: IN[0] = set()
    gen = []
    for b in blocks:
        g = set()
        for s in b: g |= du(s)[0]
        gen.append(g)
    changed = True
    while changed:
        changed = False
        for i in range(n):
            if i:
                ps = [OUT[p] for p in pred[i]]
                new_in = set.intersection(*ps) if ps else set(ALL)     # unreachable blocks: ignore
            else:
                new_in = set()
            new_out = new_in | gen[i]
            if new_in != IN[i] or new_out != OUT[i]:
                IN[i], OUT[i], changed = new_in, new_out, True
    bad = []
    for i, b in enumerate(blocks):
        if i and not pred[i]: continue
        have = set(IN[i])
        for s in b:
            d, u = du(s)
            miss = u - have
            if miss: bad.append((s, sorted(miss)))
            have |= d
    return bad


VREG = re.compile(r"\\bv(\\d+)\\b|\\bv\\[(\\d+):(\\d+)\\]")


def vregs(tok):
    out = set()
    for m in VREG.finditer(tok):
        if m.group(1) is not None: out.add(int(m.group(1)))
        else: out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def masked_prologue_writes(items):
    """AGPR copies between a label and the `s_or_b64 exec, exec, ...` of the same block whose value is read somewhere.
    Each hit is classified: the copy runs under the mask of the if-block that ends at the label, so it is harmless when
    the VGPR it parks was computed inside that block (same lanes) and WRONG when the VGPR was defined before the branch
    (lanes that skipped the block lose the value; all lanes when nobody took it)."""
    reads = set()
    for kind, s in items:
        if kind == "ins":
            op = s.split()[0]
            ops = [o.strip() for o in s.split(None, 1)[1].split(",")] if " " in s else []
            if op.startswith(READ_ONLY_FIRST): reads |= set().union(*[aregs(o) for o in ops]) if ops else set()
            elif len(ops) > 1: reads |= set().union(*[aregs(o) for o in ops[1:]])
    hits, pending, in_prologue, label_at = [], [], False, None
    for idx, (kind, s) in enumerate(items):
        if kind == "label":
            in_prologue, pending, label_at = True, [], idx
            continue
        op = s.split()[0]
        if not in_prologue: continue
        if op == "s_or_b64" and s.replace(" ", "").startswith("s_or_b64exec,exec,"):
            lab = items[label_at][1]
            # the if-block: from the nearest preceding `s_cbranch_execz <lab>` to the label, if it is straight-line code
            k, body, simple = label_at - 1, [], False
            while k >= 0:
                kk, ss = items[k]
                if kk == "label": break
                if ss.split()[0].startswith(("s_cbranch", "s_branch")):
                    simple = ss.split()[0] == "s_cbranch_execz" and ss.split()[-1] == lab
                    break
                body.append(ss); k -= 1
            defined = set()
            for ss in body:
                o = [x.strip() for x in ss.split(None, 1)[1].split(",")] if " " in ss else []
                if o and not ss.split()[0].startswith(READ_ONLY_FIRST): defined |= vregs(o[0])
            for s_, regs, src in pending:
                verdict = "unclear (not a straight-line if-block)" if not simple else \
                          ("harmless (value computed inside the block)" if src and src <= defined else "WRONG (value from before the branch)")
                hits.append((s_, regs, verdict))
            in_prologue = False
            continue
        first = s.split(None, 1)[-1].split(",")[0] if " " in s else ""
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_and_saveexec")) or "exec" in first:
            in_prologue = False
            continue
        ops = [o.strip() for o in s.split(None, 1)[1].split(",")] if " " in s else []
        if ops and not op.startswith(READ_ONLY_FIRST):
            d = aregs(ops[0]) & reads
            if d: pending.append((s, sorted(d), set().union(*[vregs(o) for o in ops[1:]]) if len(ops) > 1 else set()))
    return hits


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    paths = "--paths" in sys.argv
    path = args[0]; filt = args[1] if len(args) > 1 else ""
    funcs = parse_library(path) if path.endswith(".so") else parse(path)
    nbad = 0
    for name, items in funcs.items():
        if filt and filt not in name: continue
        if not paths:
            hits = masked_prologue_writes(items)
            wrong = [h for h in hits if not h[2].startswith("harmless")]
            if wrong: nbad += 1
            if hits and (wrong or "--all" in sys.argv):
                print("%s: %d AGPR write(s) in front of the exec restore of a join block:" % (name, len(hits)))
                for s_, regs, verdict in hits[:8]: print("    %-44s a%s  %s" % (s_, regs, verdict))
            continue
        bad = check(name, items)
        if bad:
            nbad += 1
            print("%s: %d read(s) of an AGPR that is not written on every path:" % (name, len(bad)))
            for s, miss in bad[:6]: print("    %-60s  a%s" % (s, miss))
    print("%d kernel(s) checked, %d flagged" % (len([f for f in funcs if not filt or filt in f]), nbad))
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
