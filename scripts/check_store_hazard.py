#!/usr/bin/env python3
"""ISA lint for a hazard the gfx950 code generator left open once (aux_kernels.hip, k_dense_nn_lean, round 5): a VMEM / LDS store of more
than 8 bytes reads its upper data registers a cycle after it issues, so a VALU write of one of those registers needs wait states in
between.  The compiler inserts them inside a basic block; when the store is the LAST instruction of a block and the next block starts
with such a write, nothing separated them.  usage: check_store_hazard.py [--require kernel[,kernel...]] file.s [...]
(hipcc --cuda-device-only -S gives the .s files; --require fails the run when a named kernel is in none of them: an assembly file that
did not compile, or a stale one, must not pass as "0 suspicious places" -- ADVICE r5)
Prints every place where a >8-byte store is followed, within two instructions and across labels only, by a VALU write of its data."""
import re
import sys

STORE = re.compile(r"^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34]|ds_write_b96|ds_write_b128)\s+(.*)$")
RANGE = re.compile(r"v\[(\d+):(\d+)\]")
WRITE = re.compile(r"^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))")


def data_regs(op, args):
    regs = RANGE.findall(args)
    if not regs:
        return set()
    # data operand: first vector range for buffer stores, the LAST for global / flat / ds stores with an address first
    lo, hi = (regs[0] if op.startswith("buffer") else regs[-1])
    lo, hi = int(lo), int(hi)
    return set(range(lo, hi + 1)) if hi - lo >= 2 else set()


require = []
if len(sys.argv) > 2 and sys.argv[1] == "--require":
    require = [k for k in sys.argv[2].split(",") if k]
    del sys.argv[1:3]
if len(sys.argv) < 2:
    sys.exit("usage: check_store_hazard.py [--require kernel[,kernel...]] file.s [...]")
texts = {path: open(path).read() for path in sys.argv[1:]}
empty = [p for p, t in texts.items() if ".amdhsa_kernel" not in t]
if empty:
    sys.exit("no kernel in %s: the file did not compile (or is not device assembly)" % ", ".join(empty))
missing = [k for k in require if not any(re.search(r"^_Z\w*%s\w*:" % re.escape(k), t, re.M) for t in texts.values())]
if missing:
    sys.exit("required kernel(s) not found in the assembly: %s" % ", ".join(missing))

bad = 0
for path in sys.argv[1:]:
    lines = texts[path].split("\n")
    func = "?"
    code = []   # (line number, text, is_label)
    for n, l in enumerate(lines, 1):
        t = l.split(";")[0].rstrip()
        if not t.strip():
            continue
        if re.match(r"^[A-Za-z_.$][\w.$]*:", t):
            if not t.startswith(".L"):
                func = t.split(":")[0]
            code.append((n, t, True, func))
        elif t.startswith("\t") and not t.strip().startswith("."):
            code.append((n, t, False, func))
    for i, (n, t, lab, fn) in enumerate(code):
        m = STORE.match(t)
        if not m:
            continue
        regs = data_regs(m.group(1), m.group(2))
        if not regs:
            continue
        seen, crossed = 0, False
        for (n2, t2, lab2, _) in code[i + 1:i + 8]:
            if lab2:
                crossed = True
                continue
            if t2.strip().startswith("s_nop") or t2.strip().startswith("s_waitcnt"):
                break
            w = WRITE.match(t2)
            if w and not w.group(1).startswith("v_cmp"):
                dst = set(range(int(w.group(3)), int(w.group(4)) + 1)) if w.group(3) else {int(w.group(5))}
                if dst & regs and crossed:
                    print("%s:%d %s: `%s` then (across a label) `%s`" % (path, n, fn, t.strip(), t2.strip()))
                    bad += 1
                    break
            seen += 1
            if seen >= 2:
                break
print("%d suspicious place(s)" % bad)

# Second check: k_dense_nn_ahead / _ahead2 keep their prefetch registers (v192 .. v255) out of the compiler's hands with amdgpu_num_vgpr(192) and
# count their own vmcnt.  A toolchain that stopped honouring the attribute, or spilled (scratch loads carry vmcnt waits of their own), would corrupt
# products silently: outside the ;;#ASMSTART ... ;;#ASMEND blocks no instruction of those kernels may name a VGPR >= 192 or touch scratch.
VREG = re.compile(r"\bv\[?(\d+)(?::(\d+))?\]?")
fixed_bad = 0
for path in sys.argv[1:]:
    func, inasm = None, False
    for n, l in enumerate(texts[path].split("\n"), 1):
        m = re.match(r"^(_Z\w*k_dense_nn_ahead\w*):", l)
        if m:
            func, inasm = m.group(1), False
            continue
        if func and ".Lfunc_end" in l:
            func = None
        if not func:
            continue
        if "ASMSTART" in l:
            inasm = True
            continue
        if "ASMEND" in l:
            inasm = False
            continue
        t = l.split(";")[0]
        if not t.startswith("\t") or t.strip().startswith("."):
            continue
        if "scratch_" in t:
            print("%s:%d %s: scratch access `%s`" % (path, n, func, t.strip()))
            fixed_bad += 1
        if not inasm:
            for r in VREG.finditer(t):
                if int(r.group(2) or r.group(1)) >= 192:
                    print("%s:%d %s: the compiler uses a reserved register: `%s`" % (path, n, func, t.strip()))
                    fixed_bad += 1
                    break
print("%d violation(s) of the fixed-register discipline of k_dense_nn_ahead*" % fixed_bad)
sys.exit(1 if bad or fixed_bad else 0)
