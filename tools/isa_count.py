"""Instruction mix of one kernel in a hipcc -S dump: python tools/isa_count.py file.s <mangled-name-substring>"""
import sys, re, collections
s = open(sys.argv[1]).read(); key = sys.argv[2]
names = [l.split(":")[0] for l in s.split("\n") if l.startswith("_Z") and key in l.split(":")[0] and ":" in l]
name = names[0]
i = s.index(name + ':'); j = s.index('.end_amdhsa_kernel', i)
body = s[i:j]
c = collections.Counter()
for l in body.split('\n'):
    t = l.strip().split(' ')[0]
    if not t or t.startswith(('.', ';', '_Z')) or t.endswith(':'): continue
    c[t] += 1
tot = sum(c.values())
grp = collections.Counter()
for k, v in c.items():
    g = ('mfma' if k.startswith('v_mfma') else 'v_pk' if k.startswith('v_pk_') else 'trans' if k in ('v_exp_f32','v_rcp_f32','v_rsq_f32','v_log_f32','v_sqrt_f32') else
         'valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'lds' if k.startswith('ds_') else 'vmem' if k.startswith(('global_','buffer_','flat_','scratch_')) else 'other')
    grp[g] += v
print(name, 'total', tot, dict(grp))
m = re.search(r'\.vgpr_count:\s*(\d+)', s[j:j+6000]); 
print('top:', c.most_common(28))
