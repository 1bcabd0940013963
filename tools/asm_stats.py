"""Count instruction classes per kernel in a hipcc -S dump (developer tool)."""
import re, sys, collections
txt = open(sys.argv[1]).read()
pat = sys.argv[2:] or ['']
for m in re.finditer(r'^(_Z\w+):\s*;\s*@.*?\n(.*?)s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not any(p in name for p in pat):
        continue
    c = collections.Counter(); ops = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in '.;' or line.endswith(':'):
            continue
        op = line.split()[0]
        ops[op] += 1
        if op.startswith('v_'): c['VALU'] += 1
        elif op.startswith('ds_'): c['LDS'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['VMEM'] += 1
        elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
        elif op.startswith('s_barrier'): c['barrier'] += 1
        elif op.startswith('s_load') or op.startswith('s_buffer'): c['SMEM'] += 1
        elif op.startswith('s_'): c['SALU'] += 1
    print(name[:60], dict(c))
    print('    ', ops.most_common(18))
