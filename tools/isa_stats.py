#!/usr/bin/env python3
"""Developer tool: per-kernel instruction / register statistics from a `hipcc -S --cuda-device-only` listing.
usage: isa_stats.py file.s name-substring [more substrings]"""
import re
import sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(\S+):\s*; @\1', s, re.M):
    name = m.group(1)
    if not all(k in name for k in sys.argv[2:]):
        continue
    end = s.find('.end_amdhsa_kernel', m.start())
    if end < 0:
        continue
    body = s[m.start():end]
    cnt = lambda k: len(re.findall(r'^\s+' + k, body, re.M))
    print(name)
    print('   mfma_bf16 %d  mfma_f32 %d  mfma_f64 %d  cvt_pk_bf16 %d  v_pk %d  valu(v_) %d  ds_ %d  global/buffer loads %d  scratch %d  s_waitcnt %d' % (
        cnt('v_mfma_f32_16x16x32_bf16'), cnt('v_mfma_f32_16x16x4_f32'), cnt('v_mfma_f64'), cnt('v_cvt_pk_bf16'), cnt('v_pk_'), cnt('v_'), cnt('ds_'),
        cnt('global_load') + cnt('buffer_load'), cnt('scratch_'), cnt('s_waitcnt')))
    for k in ('next_free_vgpr', 'accum_offset', 'private_segment_fixed_size', 'group_segment_fixed_size'):
        for l in re.findall(r'.*%s.*' % k, body):
            print('   ', l.strip())
