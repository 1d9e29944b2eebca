"""Audits the gfx950 ISA of libmpcombi_hip for the VGPR index mode the register simplex relies on (lp_reg.hpp: a tableau
column chosen at run time is read / written through s_set_gpr_idx_on ... s_set_gpr_idx_off).  While the mode is on EVERY
vector instruction is index-shifted, so a region must be short, closed, and free of control flow.  (During development a
build with an early `return` between an indexed read and an indexed write produced wrong verdicts and memory faults; the
source was restructured, and this check makes the property visible.)

    python tools/check_gpr_idx.py [file.s]      # without an argument: compiles ppopt_amd/csrc/mpcombi_hip.hip with --save-temps
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(text: str):
    on, bad, total, longest, func = None, [], 0, 0, None
    for i, line in enumerate(text.split('\n')):
        t = line.strip()
        if re.match(r'^_ZN3mpc\w+:', line):
            func = line.split(':')[0]
        if t.startswith('s_set_gpr_idx_on'):
            if on is not None:
                bad.append((func, i, 'nested s_set_gpr_idx_on'))
            on, total = i, total + 1
        elif t.startswith('s_set_gpr_idx_off'):
            if on is None:
                bad.append((func, i, 's_set_gpr_idx_off without on'))
            else:
                longest = max(longest, i - on)
            on = None
        elif on is not None and re.match(r'^(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc|\.LBB)', t):
            bad.append((func, i, 'control flow inside an index-mode region: ' + t))
    if on is not None:
        bad.append((func, on, 'index mode left on'))
    return total, longest, bad


def assembly() -> str:
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-c',
                               os.path.join(ROOT, 'ppopt_amd', 'csrc', 'mpcombi_hip.hip'), '-o', os.path.join(tmp, 'x.o'), '--save-temps'],
                              cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        name = [f for f in os.listdir(tmp) if f.endswith('gfx950.s')][0]
        return open(os.path.join(tmp, name)).read()


if __name__ == '__main__':
    text = open(sys.argv[1]).read() if len(sys.argv) > 1 else assembly()
    total, longest, bad = audit(text)
    print(f'{total} index-mode regions, longest {longest} lines, {len(bad)} malformed')
    for b in bad:
        print(' ', b)
    sys.exit(1 if bad or total == 0 else 0)
