"""Event skeleton (DMA / global loads+stores / scratch / MFMA / barriers / vmcnt waits) of one kernel in a hipcc -S dump.
python tools/isa_events.py file.s <mangled-name-substring>"""
import sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(":")[0] for l in s.split("\n") if l.startswith("_Z") and key in l and ":" in l]
name = names[0]
i = s.index(name + ':'); j = s.index('.end_amdhsa_kernel', i)
ev = []
for l in s[i:j].split('\n'):
    t = l.strip()
    if t.startswith('scratch_'): ev.append('SCR' + ('L' if 'load' in t else 'S'))
    elif t.startswith('v_mfma'): ev.append('M')
    elif t.startswith('global_load_lds'): ev.append('D')
    elif t.startswith('global_load'): ev.append('GL')
    elif t.startswith('global_store'): ev.append('GS')
    elif t.startswith('s_barrier'): ev.append('BAR')
    elif t.startswith('s_waitcnt') and 'vmcnt' in t: ev.append('W[' + t.split('s_waitcnt')[1].strip() + ']')
out = []; last = None; cnt = 0
for e in ev + [None]:
    if e == last: cnt += 1
    else:
        if last: out.append(f"{last}x{cnt}" if cnt > 1 else last)
        last = e; cnt = 1
print(name); print(' '.join(out))
