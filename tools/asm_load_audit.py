"""Audit of untracked (inline-asm) global loads in a hipcc -S dump: between such a load and the next inline-asm s_waitcnt vmcnt,
no compiler instruction may touch the load's destination registers (cdna_hip_programming.md 5.7 item 1: the compiler
considers them written at the asm statement and may copy / spill them before the data lands).
python tools/asm_load_audit.py file.s [kernel-name-substring]"""
import re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2] if len(sys.argv) > 2 else ""
bad_total = 0
for m in re.finditer(r'^(_Z\S+):.*?\.end_amdhsa_kernel', s, re.S | re.M):
    name = m.group(1)
    if key not in name: continue
    lines = m.group(0).split('\n')
    in_asm = False; pending = []   # list of (line_no, set(regs))
    bad = 0; loads = 0
    for n, l in enumerate(lines):
        t = l.strip()
        if t.startswith(';;#ASMSTART'): in_asm = True; continue
        if t.startswith(';;#ASMEND'): in_asm = False; continue
        if not t or t.startswith(';') or t.startswith('.'): continue
        if in_asm:
            mm = re.match(r'global_load_dwordx4 v\[(\d+):(\d+)\]', t)
            if mm and 'lds' not in t:
                pending.append((n, set(range(int(mm.group(1)), int(mm.group(2)) + 1)))); loads += 1
            elif t.startswith('s_waitcnt') and 'asm-loads-landed' in t:      # the wait statement that names the loaded registers
                pending = []
            continue
        if pending:
            regs = set()
            for a, b in re.findall(r'v\[(\d+):(\d+)\]', t): regs |= set(range(int(a), int(b) + 1))
            for a in re.findall(r'\bv(\d+)\b', t): regs.add(int(a))
            for ln, dst in pending:
                if regs & dst:
                    bad += 1
                    print(f"  {name[:70]}: line {n}: '{t[:70]}' touches v{sorted(regs & dst)} loaded at line {ln} before the wait")
    print(f"{name[:90]}: {loads} asm loads, {bad} suspicious")
    bad_total += bad
sys.exit(1 if bad_total else 0)
