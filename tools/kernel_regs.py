"""VGPR count and scratch bytes of every kernel of two builds of a translation unit, side by side (a kernel under a tight register cap can
tip into scratch on a harmless-looking change: round 6, k_x1 12 -> 104 bytes):  hipcc ... --save-temps in two directories, then
    python tools/kernel_regs.py <dir_new> <dir_old>"""
import re, glob, subprocess, sys
def parse(p):
    txt=open(p).read()
    out={}
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
        name=m.group(1); body=m.group(2)
        sc=int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1))
        vg=int(re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1))
        out[name]=(vg,sc)
    return out
new=parse(glob.glob(sys.argv[1]+'/*gfx950.s')[0]); old=parse(glob.glob(sys.argv[2]+'/*gfx950.s')[0])
def dem(n):
    try: return subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()[:80]
    except Exception: return n[:80]
for k in sorted(new):
    if k in old and new[k]!=old[k]:
        print(dem(k), 'old', old[k], 'new', new[k])
