"""Dev tool: per-basic-block instruction mix of one kernel in a hipcc -S dump.  usage: asm_blocks.py file.s <kernel-substring> [min_len]"""
import re, sys, collections
txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]; minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 60
start = next(i for i, l in enumerate(txt) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith('E') and ':' in l)
end = next(i for i in range(start, len(txt)) if 's_endpgm' in txt[i])
blocks = []; cur = ['entry', []]
for l in txt[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith((';', '.', '//')):
        if re.match(r'^\.LBB\d+_\d+:', s): blocks.append(cur); cur = [s.split(':')[0], []]
        continue
    cur[1].append(s.split()[0])
blocks.append(cur)
print('kernel lines', end - start, 'blocks', len(blocks))
for name, ins in blocks:
    if len(ins) < minlen: continue
    c = collections.Counter(ins)
    f64 = c['v_fma_f64'] + c['v_mul_f64'] + c['v_add_f64'] + c['v_max_f64'] + c['v_min_f64']
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print('%-10s n=%4d valu=%4d [fma64 %d mul64 %d add64 %d max/min64 %d] dpp %d mov %d cndmask %d cmp %d salu %d lds %d vmem %d' % (
        name, len(ins), valu, c['v_fma_f64'], c['v_mul_f64'], c['v_add_f64'], c['v_max_f64'] + c['v_min_f64'], c['v_mov_b32_dpp'],
        c['v_mov_b32_e32'] + c['v_mov_b64_e32'] + c.get('v_accvgpr_write_b32', 0) + c.get('v_accvgpr_read_b32', 0),
        c['v_cndmask_b32_e32'] + c['v_cndmask_b32_e64'], sum(v for k, v in c.items() if k.startswith('v_cmp')),
        sum(v for k, v in c.items() if k.startswith('s_')), sum(v for k, v in c.items() if k.startswith('ds_')),
        sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))))
    rest = collections.Counter({k: v for k, v in c.items() if k.startswith('v_') and k not in ('v_fma_f64', 'v_mul_f64', 'v_add_f64', 'v_mov_b32_dpp')})
    print('           ', rest.most_common(12))
