import re, sys
s = open('/tmp/conv3x3v-hip-amdgcn-amd-amdhsa-gfx950.s').read()
names = sys.argv[1:] or ['_Z15conv3x3v_kernelILi16ELi2ELi2ELi0ELi1ELi0EEv9Conv3Args', '_Z15conv3x3v_kernelILi32ELi4ELi1ELi0ELi1ELi0EEv9Conv3Args']
for nm in names:
    full = s.split(nm + ':')[1]
    f = full.split('.Lfunc_end')[0]
    vg = re.search(r'; NumVgprs: (\d+)', full); ag = re.search(r'; NumAgprs: (\d+)', full); sp = re.search(r'; ScratchSize: (\d+)', full)
    print(nm, 'V', vg.group(1), 'A', ag.group(1), 'scratch', sp.group(1))
    lines = [l.rstrip() for l in f.split('\n') if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
    bar = [i for i,l in enumerate(lines) if 's_barrier' in l]
    start = bar[0]
    out = []
    for l in lines[start:start+700]:
        t = l.split(); op = t[0]
        if op.startswith('v_mfma'): out.append('M')
        elif op == 'ds_read_b128': out.append('d')
        elif op.startswith('global_load_lds'): out.append('P')
        elif op.startswith('global_load'): out.append('G')
        elif op == 's_waitcnt': out.append('[' + ' '.join(t[1:]) + ']')
        elif op == 's_barrier': out.append('|BAR|')
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): out.append('<BR>')
        elif op == 's_nop': out.append('n')
        elif op.startswith('v_accvgpr'): out.append('A')
        elif op.startswith('scratch'): out.append('#')
        elif op.startswith('v_'): out.append('v')
        elif op.startswith('s_'): out.append('s')
        else: out.append('?'+op)
    print(''.join(out))
