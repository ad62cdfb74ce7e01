"""Static check of the persistent conv kernel's ISA: prints, per kernel, the order of vector-memory operations, vmcnt waits and barriers
(run-length compressed).  The partial waits in agf_conv2d_pipe.hip are only correct if every wave issues exactly the operations the
constants assume -- no spills (scratch_*), no compiler-inserted vmcnt(0) inside the tile loop.   python tools/check_pipe_isa.py [regex]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'animeface_amd', 'csrc', 'agf_conv2d_pipe.hip')
asm = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=fast', '-munsafe-fp-atomics', '-x', 'hip',
                      '--cuda-device-only', '-S', src, '-o', '-'], capture_output=True, text=True).stdout
filt = re.compile(sys.argv[1] if len(sys.argv) > 1 else '.')
cur, ops = None, []
def flush():
    if cur and filt.search(cur):
        out, last, n = [], None, 0
        for o in ops + [None]:
            if o == last: n += 1; continue
            if last is not None: out.append(f'{last}x{n}' if n > 1 else last)
            last, n = o, 1
        print(cur); print('   ', ' '.join(out)); print('    scratch ops:', sum(o.startswith('scratch') for o in ops))
for line in asm.splitlines():
    m = re.match(r'^(_Z\w+):', line)
    if m: flush(); cur, ops = m.group(1), []; continue
    t = line.strip()
    if t.startswith('.amdhsa_kernel'): flush(); cur = None
    if cur is None: continue
    if t.startswith('buffer_load') and ' lds' in t: ops.append('DMA')
    elif t.startswith('buffer_load'): ops.append('LD')
    elif t.startswith('buffer_store'): ops.append('ST')
    elif t.startswith('scratch_'): ops.append('scratch')
    elif t.startswith('global_') or t.startswith('flat_'): ops.append(t.split()[0])
    elif t.startswith('s_barrier'): ops.append('|BAR|')
    elif t.startswith('s_waitcnt') and 'vmcnt' in t: ops.append('w' + re.search(r'vmcnt\((\d+)\)', t).group(1))
    elif t.startswith('s_cbranch'): ops.append('br')
    elif re.match(r'^\.LBB\d+_\d+:', t): ops.append('L:')
    elif t.startswith('v_mfma'): ops.append('M')
