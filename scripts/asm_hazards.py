#!/usr/bin/env python3
"""Static scan of the assembled intersect kernel for the gfx940-family hazards the assembler does not pad (LLVM's GCNHazardRecognizer inserts the
s_nops for compiled code; hand-written code has to carry them).  Works on the disassembly of a code object as a control-flow graph: every rule is
evaluated along EVERY path, forwards through both outcomes of a conditional branch, backwards through every predecessor of a label (taken branches
and loop back-edges included):
    asm_hazards.py [build/asm/pt_extend_hsaco/pt_extend_s16.hsaco ...]
Wait-state rules (wait states = instructions issued in between on the path; s_nop N counts N + 1):
  A  VALU writes an SGPR / VCC        -> VALU reads it as an operand or mask            2
  B  VALU writes an SGPR / VCC        -> v_readlane / v_writelane lane select            4
  C  VALU writes VCC                  -> v_div_fmas                                      4
  D  VALU writes an SGPR              -> VMEM / DS / FLAT reads it                       5
  E  VALU writes a VGPR               -> v_readfirstlane / v_readlane reads it           1
  F  transcendental VALU              -> VALU reads its result                           1
  G  VMEM / DS store of > 8 B of data -> VALU writes one of the data registers           1   (2 with an SGPR offset; none here)
  H  VALU writes EXEC (v_cmpx)        -> v_readlane / v_readfirstlane / v_writelane      4
Memory-return rules (forward walk from every load, no depth limit: a walk ends at a covering wait, at s_endpgm or where it has been before):
  S  a scalar load (s_load / s_memtime) whose destination registers are read or written, or whose wave ends, before an s_waitcnt lgkmcnt(0):
     scalar loads return out of order and nothing interlocks them — a register reused early is overwritten when the data lands.
  V  a vector-memory load (vmcnt) or an LDS read (lgkmcnt) whose destination VGPRs are read, or written by anything but another load, before an
     s_waitcnt that covers it.  Loads of one counter return in issue order, so a load with k younger operations of its counter issued behind it is
     complete after s_waitcnt <counter>(N) with N <= k: the walk carries k (the minimum over the paths that meet).  Another load into the same
     registers is not a touch (the two arms of a fetch write disjoint lanes of the same registers and share one wait).
  X  the scan itself could not follow a path (a branch whose target is not an instruction of the object): reported, never skipped.
"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_exp_", "v_sin_", "v_cos_", "v_rcp_iflag")
MAX_NEED = 5          # the longest wait-state requirement above


def regs(tok):
    """'s[4:5]' -> {'s4','s5'}, 'v12' -> {'v12'}, 'vcc' -> {'vcc_lo','vcc_hi'}, 'exec' -> {...}"""
    tok = tok.strip()
    m = re.fullmatch(r"([sva])\[(\d+):(\d+)\]", tok)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"-?\|?([sva])(\d+)\|?", tok)
    if m:
        return {f"{m.group(1)}{m.group(2)}"}
    if tok in ("vcc", "exec"):
        return {tok + "_lo", tok + "_hi"}
    if tok in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi", "m0"):
        return {tok}
    return set()


def parse(line):
    line = line.split("//")[0].strip()
    if not line or line.endswith(":") or line.startswith("<"):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    ops = []
    if len(parts) > 1:
        for t in re.split(r",\s*", parts[1]):
            t = t.split()[0] if t.split() else t           # drop modifiers ("offset:16", "op_sel:[...]")
            ops.append(t)
    return op, ops


def classify(op, ops):
    """-> (kind, writes, reads) with register sets; kind in valu / salu / vmem / ds / smem / other"""
    w, r = set(), set()
    if op.startswith("v_"):
        kind = "valu"
        if op.startswith("v_cmpx"):
            w |= {"exec_lo", "exec_hi"}
            if op.endswith("_e32") or len(ops) == 2:
                w |= {"vcc_lo", "vcc_hi"}
            srcs = ops[1:]                                  # (the first operand is the destination, vcc in the e32 form)
            w |= regs(ops[0])
        elif op.startswith("v_cmp"):
            w |= regs(ops[0]); srcs = ops[1:]
        elif op.startswith(("v_readfirstlane", "v_readlane")):
            w |= regs(ops[0]); srcs = ops[1:]
        elif op.startswith(("v_div_scale", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_mad_u64_u32", "v_mad_i64_i32")):
            w |= regs(ops[0]) | regs(ops[1]); srcs = ops[2:]
        else:
            w |= regs(ops[0]) if ops else set(); srcs = ops[1:]
        for t in srcs:
            r |= regs(t)
        if op.startswith(("v_fmac_", "v_mac_", "v_pk_fmac", "v_dot2c", "v_dot4c", "v_dot8c")):
            r |= w                                          # the destination is also the accumulator
        if op.startswith(("v_cndmask_b32_e32", "v_addc_co_u32_e32", "v_subb_co_u32_e32", "v_div_fmas")) or (op.startswith("v_cndmask") and len(ops) == 3):
            r |= {"vcc_lo", "vcc_hi"}
        return kind, w, r
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        kind = "vmem"
        if "load" in op or "atomic" in op and len(ops) >= 4:
            w |= regs(ops[0]); srcs = ops[1:]
        else:
            srcs = ops
        for t in srcs:
            r |= regs(t)
        return kind, w, r
    if op.startswith("ds_"):
        kind = "ds"
        if "read" in op or "rtn" in op:
            w |= regs(ops[0]); srcs = ops[1:]
        else:
            srcs = ops
        for t in srcs:
            r |= regs(t)
        return kind, w, r
    if op.startswith("s_load") or op.startswith("s_memtime") or op.startswith("s_buffer"):
        w |= regs(ops[0])
        for t in ops[1:]:
            r |= regs(t)
        return "smem", w, r
    if op.startswith("s_"):
        if op.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_setprio", "s_bitcmp", "s_sleep")):
            for t in ops:
                r |= regs(t)
            return "salu", w, r
        w |= regs(ops[0]) if ops else set()
        for t in ops[1:]:
            r |= regs(t)
        if "saveexec" in op:
            w |= {"exec_lo", "exec_hi"}; r |= {"exec_lo", "exec_hi"}
        return "salu", w, r
    return "other", w, r


def store_data(op, ops):
    """data registers of a store wider than 8 B"""
    if op.startswith(("global_store_dwordx3", "global_store_dwordx4", "flat_store_dwordx3", "flat_store_dwordx4")):
        return regs(ops[1])
    if op.startswith(("ds_write_b96", "ds_write_b128")):
        return regs(ops[1])
    if op.startswith(("ds_write2_b64", "ds_write2st64_b64")):
        return regs(ops[1]) | regs(ops[2])
    return set()


def waitcnt(txt):
    """'s_waitcnt vmcnt(0) lgkmcnt(1)' -> {'vmcnt': 0, 'lgkmcnt': 1}; a counter that is not named is not waited for"""
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)", txt)}


class Program:
    def __init__(self, path):
        text = subprocess.check_output([OBJDUMP, "-d", path], text=True)
        self.ins, self.raw = [], []
        for l in text.splitlines():
            if "//" not in l or not l.startswith("\t"):
                continue
            addr = l.split("//")[1].split(":")[0].strip()
            p = parse(l)
            if p:
                self.ins.append((addr, l.split("//")[0].strip(), *p)); self.raw.append(l)
        self.index = {a: i for i, (a, *_r) in enumerate(self.ins)}
        self.cls = [classify(op, ops) for (_a, _t, op, ops) in self.ins]
        n = len(self.ins)
        self.succ = [[] for _ in range(n)]
        self.pred = [[] for _ in range(n)]
        self.unresolved = []                                # branches whose target is not an instruction of this object
        for i, (addr, txt, op, ops) in enumerate(self.ins):
            nxt = []
            if op.startswith("s_endpgm"):
                pass
            elif op.startswith("s_branch"):
                t = self.target(i)
                if t is None:
                    self.unresolved.append(i)
                else:
                    nxt.append(t)
            elif op.startswith("s_cbranch"):
                t = self.target(i)
                if t is None:
                    self.unresolved.append(i)
                else:
                    nxt.append(t)
                if i + 1 < n:
                    nxt.append(i + 1)
            elif op.startswith(("s_setpc", "s_swappc")):
                self.unresolved.append(i)
            elif i + 1 < n:
                nxt.append(i + 1)
            for t in nxt:
                if t not in self.succ[i]:
                    self.succ[i].append(t); self.pred[t].append(i)

    def target(self, i):                                    # SOPP branch: target = address + 4 + 4 * simm16 (the low half of the instruction word)
        a, word = self.raw[i].split("//")[1].split(":")
        simm = int(word.split()[0], 16) & 0xffff
        simm -= 0x10000 if simm & 0x8000 else 0
        return self.index.get("%012X" % (int(a.strip(), 16) + 4 + 4 * simm))

    def states_of(self, j):                                 # wait states instruction j provides to what follows it
        _a, _t, op, ops = self.ins[j]
        return (int(ops[0], 0) + 1) if op == "s_nop" and ops else 1


# Findings the scan cannot clear by itself because it does not track exec masks: (rule, producer regex, consumer regex, why it is not a hazard).
# A waiver is applied only by scan(..., waive=True); what it suppressed is returned beside the findings, and the test pins the exact set.
WAIVERS = []      # (round 5: the one former entry — the node step's pop into vCur against the triangle step's advance of vCur — is gone from the kernel: pops land in vPop)


def scan(path, waive=False):
    """-> (instructions, findings) or, with waive=True, (instructions, findings, waived)"""
    n, found = _scan(path)
    if not waive:
        return n, found
    kept, waived = [], []
    for f in found:
        w = next((k for k, (rule, prod, cons, _why) in enumerate(WAIVERS) if f[0] == rule and re.fullmatch(prod, f[4]) and re.fullmatch(cons, f[6])), None)
        (waived if w is not None else kept).append(f if w is None else (w, f))
    return n, kept, waived


def _scan(path):
    P = Program(path)
    ins = P.ins
    found = []
    for i in P.unresolved:
        found.append(("X", 0, 0, ins[i][0], ins[i][1], ins[i][0], "branch target is not an instruction of this object: the paths behind it were not scanned", []))

    # ---- wait-state rules: look back from every instruction along every predecessor path, up to MAX_NEED wait states
    def pair(rule_hits, i, j, states):
        addr, txt, op, ops = ins[i]
        kind, w, r = P.cls[i]
        a2, t2, op2, ops2 = ins[j]
        k2, w2, r2 = P.cls[j]

        def hit(rule, need, regs_):
            if regs_ and states < need:
                rule_hits.add((rule, need, states, a2, t2, addr, txt, tuple(sorted(regs_))))
        sg = {x for x in w2 if x[0] == "s" or x.startswith("vcc")} if k2 == "valu" else set()
        if kind == "valu":
            lane_sel = op.startswith(("v_readlane", "v_writelane"))
            if lane_sel and ops:
                hit("B", 4, sg & regs(ops[-1]))
            hit("A", 2, sg & r)
            if op.startswith("v_div_fmas"):
                hit("C", 4, sg & {"vcc_lo", "vcc_hi"})
            if op.startswith(("v_readfirstlane", "v_readlane")) and k2 == "valu":
                hit("E", 1, {x for x in w2 if x[0] == "v"} & r)
            if k2 == "valu" and op2.startswith(TRANS):
                hit("F", 1, {x for x in w2 if x[0] == "v"} & r)
            if k2 in ("vmem", "ds"):
                hit("G", 1, store_data(op2, ops2) & w)
            if op.startswith(("v_readfirstlane", "v_readlane", "v_writelane")) and k2 == "valu" and op2.startswith("v_cmpx"):
                hit("H", 4, {"exec"})
        if kind in ("vmem", "ds"):
            hit("D", 5, {x for x in sg if x[0] == "s"} & r)

    hits = set()
    for i in range(len(ins)):
        kind = P.cls[i][0]
        if kind not in ("valu", "vmem", "ds"):
            continue
        seen = set()
        work = [(j, 0) for j in P.pred[i]]
        while work:
            j, states = work.pop()
            if (j, states) in seen or states >= MAX_NEED:
                continue
            seen.add((j, states))
            pair(hits, i, j, states)
            ns = states + P.states_of(j)
            for q in P.pred[j]:
                work.append((q, ns))
    found += [(*h[:7], list(h[7])) for h in sorted(hits)]

    # ---- S: from every scalar load to the first full lgkmcnt wait on every path
    for i, (addr, txt, op, ops) in enumerate(ins):
        if P.cls[i][0] != "smem":
            continue
        dest = regs(ops[0])
        seen, work = set(), list(P.succ[i])
        if not P.succ[i]:
            found.append(("S", 0, 0, addr, txt, addr, "(no successor)", sorted(dest)))
        while work:
            j = work.pop()
            if j in seen:
                continue
            seen.add(j)
            a2, t2, op2, ops2 = ins[j]
            if op2 == "s_waitcnt" and waitcnt(t2).get("lgkmcnt") == 0:
                continue
            if op2.startswith("s_endpgm"):
                continue                                    # (s_endpgm waits for everything outstanding)
            k2, w2, r2 = P.cls[j]
            if (w2 | r2) & dest and not (k2 == "smem" and not (r2 & dest)):
                found.append(("S", 0, 0, addr, txt, a2, t2, sorted((w2 | r2) & dest)))
                continue
            work += P.succ[j]

    # ---- V: from every vector-memory load / LDS read to a wait that covers it, on every path
    for i, (addr, txt, op, ops) in enumerate(ins):
        kind, w, _r = P.cls[i]
        if kind not in ("vmem", "ds") or not w:
            continue
        counter = "vmcnt" if kind == "vmem" else "lgkmcnt"
        dest = {x for x in w if x[0] in "va"}
        if not dest:
            continue
        best = {}                                           # instruction -> smallest k (younger operations of the counter) it has been reached with
        work = [(j, 0) for j in P.succ[i]]
        while work:
            j, k = work.pop()
            if j in best and best[j] <= k:
                continue
            best[j] = k
            a2, t2, op2, ops2 = ins[j]
            if op2 == "s_waitcnt":
                n = waitcnt(t2).get(counter)
                if n is not None and n <= k:
                    continue                                # covered on this path
            if op2.startswith("s_endpgm"):
                continue
            k2, w2, r2 = P.cls[j]
            is_load = k2 in ("vmem", "ds") and bool(w2)
            touched = (r2 & dest) | (set() if is_load else (w2 & dest))
            if touched and j != i:
                found.append(("V", 0, k, addr, txt, a2, t2, sorted(touched)))
                continue
            k2n = min(k + 1, 64) if k2 == kind else k       # (every operation of the kind counts: loads, stores, atomics; LDS-DMA is not used here)
            for q in P.succ[j]:
                work.append((q, k2n))
    return len(ins), found


if __name__ == "__main__":
    paths = sys.argv[1:] or ["build/asm/pt_extend_hsaco/pt_extend_s16.hsaco"]
    bad = 0
    for p in paths:
        n, found, waived = scan(p, waive=True)
        print(f"{p}: {n} instructions, {len(found)} finding(s), {len(waived)} waived (lane-disjoint, see WAIVERS)")
        for rule, need, have, a2, t2, a, t, rg in found:
            if rule in ("S", "V", "X"):
                print(f"  {rule}: {a2}  {t2}   ->   {a}  {t}   [{', '.join(rg)}]" + (f"  ({have} younger operation(s) of the counter on this path)" if rule == "V" else ""))
            else:
                print(f"  {rule}: needs {need} wait state(s), has {have}: {a2}  {t2}   ->   {a}  {t}   [{', '.join(rg)}]")
        bad += len(found)
    sys.exit(1 if bad else 0)
