#!/usr/bin/env python3
"""Static scan of the assembled intersect kernel for the gfx940-family hazards the assembler does not pad (LLVM's GCNHazardRecognizer inserts the
s_nops for compiled code; hand-written code has to carry them).  Works on the disassembly of a code object, in address order (fall-through paths):
    asm_hazards.py [build/asm/pt_extend_hsaco/pt_extend_s16.hsaco ...]
Checked (wait states = instructions issued in between; s_nop N counts N + 1):
  A  VALU writes an SGPR / VCC        -> VALU reads it as an operand or mask            2
  B  VALU writes an SGPR / VCC        -> v_readlane / v_writelane lane select            4
  C  VALU writes VCC                  -> v_div_fmas                                      4
  D  VALU writes an SGPR              -> VMEM / DS / FLAT reads it                       5
  E  VALU writes a VGPR               -> v_readfirstlane / v_readlane reads it           1
  F  transcendental VALU              -> VALU reads its result                           1
  G  VMEM / DS store of > 8 B of data -> VALU writes one of the data registers           1   (2 with an SGPR offset; none here)
  H  VALU writes EXEC (v_cmpx)        -> v_readlane / v_readfirstlane / v_writelane      4
and, along every path (branches followed both ways):
  S  a scalar load (s_load / s_memtime) whose destination registers are read or written, or whose wave ends, before an s_waitcnt lgkmcnt(0):
     scalar loads return out of order and nothing interlocks them — a register reused early is overwritten when the data lands.
"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_exp_", "v_sin_", "v_cos_", "v_rcp_iflag")


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
        if op.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_setprio", "s_bitcmp")):
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


def scan(path):
    text = subprocess.check_output([OBJDUMP, "-d", path], text=True)
    ins, raw = [], []
    for l in text.splitlines():
        if "//" not in l or not l.startswith("\t"):
            continue
        addr = l.split("//")[1].split(":")[0].strip()
        p = parse(l)
        if p:
            ins.append((addr, l.split("//")[0].strip(), *p)); raw.append(l)
    found = []
    for i, (addr, txt, op, ops) in enumerate(ins):
        kind, w, r = classify(op, ops)
        # look back
        states = 0
        for j in range(i - 1, max(i - 8, -1), -1):
            a2, t2, op2, ops2 = ins[j]
            k2, w2, r2 = classify(op2, ops2)
            def hit(rule, need, regs_):
                if regs_ and states < need:
                    found.append((rule, need, states, a2, t2, addr, txt, sorted(regs_)))
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
            states += (int(ops2[0], 0) + 1) if op2 == "s_nop" and ops2 else 1
            if op2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break                                   # (what precedes a branch on the taken path is not in address order)
    # S: path-sensitive walk from every scalar load to the first full lgkmcnt wait
    index = {a: i for i, (a, *_rest) in enumerate(ins)}

    def target(i):                                       # SOPP branch: target = address + 4 + 4 * simm16 (the low half of the instruction word)
        a, word = raw[i].split("//")[1].split(":")
        simm = int(word.split()[0], 16) & 0xffff
        simm -= 0x10000 if simm & 0x8000 else 0
        return index.get("%012X" % (int(a.strip(), 16) + 4 + 4 * simm))

    for i, (addr, txt, op, ops) in enumerate(ins):
        if not (op.startswith("s_load") or op.startswith("s_memtime") or op.startswith("s_buffer_load")):
            continue
        dest = regs(ops[0])
        seen, work = set(), [(i + 1, 0)]
        while work:
            j, depth = work.pop()
            while j is not None and j < len(ins) and j not in seen and depth < 400:
                seen.add(j); depth += 1
                a2, t2, op2, ops2 = ins[j]
                if op2 == "s_waitcnt" and ("lgkmcnt(0)" in t2):
                    break
                k2, w2, r2 = classify(op2, ops2)
                if op2.startswith("s_endpgm"):
                    break                                # (s_endpgm waits for everything outstanding)
                if (w2 | r2) & dest and not (k2 == "smem" and not (r2 & dest)):
                    found.append(("S", 0, 0, addr, txt, a2, t2, sorted((w2 | r2) & dest)))
                    break
                if op2.startswith("s_branch"):
                    j = target(j); continue
                if op2.startswith("s_cbranch"):
                    tj = target(j)
                    if tj is not None:
                        work.append((tj, depth))
                j += 1
    return len(ins), found


if __name__ == "__main__":
    paths = sys.argv[1:] or ["build/asm/pt_extend_hsaco/pt_extend_s16.hsaco"]
    bad = 0
    for p in paths:
        n, found = scan(p)
        print(f"{p}: {n} instructions, {len(found)} hazard(s)")
        for rule, need, have, a2, t2, a, t, rg in found:
            print(f"  {rule}: needs {need} wait state(s), has {have}: {a2}  {t2}   ->   {a}  {t}   [{', '.join(rg)}]")
        bad += len(found)
    sys.exit(1 if bad else 0)
