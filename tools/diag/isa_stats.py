"""Resource usage and instruction mix of selected kernels in a hipcc -S listing: python tools/diag/isa_stats.py file.s pattern"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [n for n in re.findall(r'^(_Z\w+):', s, re.M) if re.search(pat, n)]
for n in names:
    i = s.index('\n' + n + ':'); j = s.index('.Lfunc_end', i); lines = s[i:j].split('\n')
    k = s.index('.amdhsa_kernel ' + n); blk = s[k:k + 5000]
    g = lambda key: (re.search(r'\.amdhsa_' + key + r'\s+(\S+)', blk) or [None, None])[1]
    cnt = lambda w: sum(w in l for l in lines)
    print(n[:60], 'vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'scratch', g('private_segment_fixed_size'), '| lines', len(lines), 'mfma', cnt('v_mfma'),
          'ds_read', cnt('ds_read'), 'ds_write', cnt('ds_write'), 'scratch_ops', cnt('scratch_'), 'barrier', cnt('s_barrier'), 'dma', cnt('global_load_lds'),
          'vmcnt0', sum(bool(re.search(r'vmcnt\(0\)', l)) for l in lines), 'v_mov_b64', cnt('v_mov_b64'), 'readlane', cnt('v_readlane'))
