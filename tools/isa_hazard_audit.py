"""Static check of a `hipcc -S` dump for instruction forms this project must not ship.

1. Packed-fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) with an `op_sel:` bit set, i.e. the LOW
   result selecting the HIGH dword of a source pair.  On MI355X that form intermittently drops its low-half product in lanes
   48..63 when the workgroup shares its CU with another kernel's waves (round 4, DESIGN.md section 2a: hipcc emitted it for
   one tap of in_conv1_kernel<32>; A/B of nothing but the operand selection: 0 vs 367 corrupted of 600 concurrent forwards).
   `op_sel_hi` (the HIGH result selecting a LOW dword) is the form used everywhere else and is not affected.  The measured map
   (tools/pk_opsel_map.py, profiles/r04_pk_opsel_map.txt): of the 96 (op, op_sel, op_sel_hi) combinations exactly those with
   op_sel[0] = 0 and op_sel[1] = 1 fail beside fp16 MFMAs; the audit bans every op_sel bit, which is simpler and costs nothing
   (hipcc reaches for op_sel only when a scalar operand happens to sit in the high half of an aligned pair).
2. Encoded `s_waitcnt vmcnt(N)` immediates above 63 (6-bit field) -- the hand-counted LDS-DMA protocol.

python tools/isa_hazard_audit.py file.s [...]   -> exit status 1 if anything is found."""
import re
import sys


def audit(path):
    bad = []
    kernel = "?"
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^(_Z\S+):", t)
        if m:
            kernel = m.group(1)
            continue
        if re.match(r"v_pk_(fma|mul|add)_f32|v_pk_mov_b32", t):
            sel = re.search(r"op_sel:\[([0-9,]+)\]", t)
            if sel and "1" in sel.group(1):
                bad.append((path, n, kernel, "packed-fp32 op with op_sel high-select: " + t))
        w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", t)
        if w and int(w.group(1)) > 63:
            bad.append((path, n, kernel, "vmcnt immediate beyond the 6-bit field: " + t))
    return bad


if __name__ == "__main__":
    found = []
    for p in sys.argv[1:]:
        found += audit(p)
    for path, n, kernel, msg in found:
        print(f"{path}:{n}: {kernel[:80]}: {msg}")
    print(f"{len(sys.argv) - 1} file(s), {len(found)} finding(s)")
    sys.exit(1 if found else 0)
