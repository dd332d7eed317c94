"""Run-length pattern of instruction classes (M mfma, V valu, D ds_read, L lds-dma, w waitcnt, n nop, p setprio, B barrier, s salu, G global) in the
densest MFMA window of one kernel of a listing.  usage: python scratch/isa_pattern.py <listing.s> <mangled-name regex> [window]"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]; win = int(sys.argv[3]) if len(sys.argv) > 3 else 700
starts = [k for k, l in enumerate(lines) if re.match(r'^_ZN.*' + pat + r'.*:\s*(;.*)?$', l)]
i = starts[0]; j = i
while not lines[j].startswith('.Lfunc_end'): j += 1
def cls(l):
    l = l.strip()
    if not l or l[0] in ';.' or l.endswith(':'): return None
    op = l.split()[0]
    for pre, c in (('v_mfma', 'M'), ('ds_read', 'D'), ('ds_', 'd'), ('global_load_lds', 'L'), ('global_', 'G'), ('buffer_', 'G'), ('scratch_', 'X'), ('s_waitcnt', 'w'), ('s_barrier', 'B'),
                   ('s_nop', 'n'), ('s_setprio', 'p'), ('s_cbranch', 'J'), ('s_branch', 'J'), ('s_', 's'), ('v_', 'V')):
        if op.startswith(pre): return c
    return '?'
seq = ''.join(c for c in (cls(l) for l in lines[i:j]) if c)
def rle(s):
    out = []; p = s[0]; n = 1
    for c in s[1:]:
        if c == p: n += 1
        else: out.append(p + (str(n) if n > 1 else '')); p = c; n = 1
    out.append(p + (str(n) if n > 1 else '')); return ' '.join(out)
best = max(range(0, max(1, len(seq) - win), 10), key=lambda k: seq[k:k + win].count('M'))
print(f"{len(seq)} instructions, {seq.count('M')} MFMA, {seq.count('V')} VALU, {seq.count('X')} scratch;  densest {win}-instruction window:")
print(rle(seq[best:best + win]))
