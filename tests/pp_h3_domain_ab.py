import re
def rows(path):
    out=[]; cur=None
    for l in open(path):
        l=l.strip()
        if l.startswith('=='): cur=l[3:]
        m=re.search(r'S=50 B=(\d+): [\d.]+ ms total, ([\d.]+) ms/step', l)
        if m and cur: out.append((cur, float(m.group(2)))); cur=None
    return out
def table(path):
    d={}
    for k,v in rows(path): d.setdefault(k,[]).append(v)
    return d
print("""What giving the split-f16 arithmetic of conv_gemm (H3) a DOMAIN costs -- the A/B series of round 5.
Every block is ONE gpurun call = one box: tests/gpu_run.sh TAG ab:LIBS@B (tests/gpu_probe.py --quick: the 50-step DDIM loop of the shipped U-Net,
z = 512, eager launches), the in-tree library ("new") and saved builds alternating, twice.  r4base = the round-4 library (commit 7aba5f3).
Boxes of the pool differ by up to 25 %: compare inside a block only.  ms per DDIM step, mean of the two runs; (+x %) against r4base of the same block.
""")
blocks=[('r5c1','gpurun_out/r5c1_ab_r4base.log',"v1: ask first (all transforms -> lane max -> wave-wide question -> all splits), two-sided band trigger, scale starts at 1, slow path = 6 ds_bpermute steps, weight scale from the set's device word (scalar load)"),
        ('r5c2','gpurun_out/r5c2_ab_r4base_nowload_nodyn.log',"v1 with the weight-scale word as a VECTOR load (new); nodyn = v1 with the dynamic activation scale compiled out (-DMUGD_H3_DYN=0); nowload = NOT a valid arm (unit 1 / S_w on scaled weights: activations 2^13 too large, every chunk off band)"),
        ('r5c3','gpurun_out/r5c3_ab_r4base_nomix.log',"v2: initial scale 2^8 (O(1) data mid-band), slow path in the exponent domain (DPP row rotations + v_readlane + scalar ALU), 1 / S_w by value for sets packed at compile time; new = with the 3-instruction v_fma_mix split, nomix = compiler-made split"),
        ('r5c4','gpurun_out/r5c4_ab_r4base_nomix.log',"v3 (new): park speculatively, check afterwards; the pipelined loop bails out to slow_tail (nomix = v2 without mix, as above)"),
        ('r5c6','gpurun_out/r5c6_ab_r4base.log',"v3 + the refill loads issued between the speculative stores and the check's branch"),
        ('r5c8','gpurun_out/r5c8_ab_r4base.log',"v4: STATIC scales for normalised operands (no per-chunk work in the XFK >= 1 loops), single-branch check, unscale merged into the cross-term fold"),
        ('r5c11','gpurun_out/r5c11_ab_r4base_nomix2.log',"v4 (new, with v_fma_mix) against v4 built with -DMUGD_H3_MIX=0 (nomix2)"),
        ('r5c12','gpurun_out/r5c12_ab_r4base_nodyn2.log',"v4 against v4 with the dynamic check compiled out (nodyn2: raw operands unprotected) -- the floor of the per-launch part"),
        ('r5c13','gpurun_out/r5c13_ab_r4base.log',"v4 + no accumulator rescale on a wave's first segment"),
        ('r5c14','gpurun_out/r5c14_ab_r4base.log',"v5: v4 + the scale state in scalar registers"),
        ('r5c16','gpurun_out/r5c16_ab_r4base.log',"v5 + lane max before the refill loads, short-circuit check, window quad pinned (ds_write_b128 back)"),
        ('r5c19','gpurun_out/r5c19_ab_r4base.log',"v6: TRACK -- raw operands at the fixed scale 2^8 with a running maximum, one check per slice, tile redo in the careful mode (register footprints back to round 4's)"),
        ('r5c21','gpurun_out/r5c21_ab_r4base.log',"v6 with four independent running maxima"),
        ('r5c23','gpurun_out/r5c23_ab_r4base.log',"v6 + sched_barrier in front of the refill loads (ineffective: the maxima, pure arithmetic to the IR, sank below it) -- a slow box"),
        ('r5c24','gpurun_out/r5c24_ab_r4base.log',"v7 (FINAL): v6 + the maxima pinned in front of the barrier -- the refill loads no longer land in spare registers behind an s_waitcnt vmcnt(0) inside the pipelined loop")]
for tag,path,desc in blocks:
    try: d=table(path)
    except OSError: continue
    print('--- %s: %s'%(tag,desc))
    for B in ('4','8','16'):
        base=[v for k,vs in d.items() if k.startswith('B=%s '%B) and 'r4base' in k for v in vs]
        if not base: continue
        b=sum(base)/len(base)
        line='  batch %2s: r4base %.3f'%(B,b)
        for k,vs in d.items():
            if k.startswith('B=%s '%B) and 'r4base' not in k:
                m=sum(vs)/len(vs)
                line+=' | %s %.3f (%+.1f %%)'%(k.split('lib=')[1],m,100*(m/b-1))
        print(line)
    print()
print("""Per-shape view of the same question (tests/gpu_convbench.py --compare: the launch the host rules pick for each hot shape of the shipped U-Net at batch 4,
cold weights, 200 back-to-back launches; us per launch, one box per table).

r5c22 -- what the raw-operand tracking costs and why (nodyn3 = dynamic handling compiled out; trk1 = running maxima only; trk2 = verdict + redo pass only):""")
import re as _re
def _load(p):
    d={}
    try:
        for l in open(p):
            m=_re.match(r"(.+?)\s+([\d.]+) us$", l.rstrip())
            if m: d[m.group(1).strip()]=float(m.group(2))
    except OSError: pass
    return d
L={k:_load('gpurun_out/r5c22_cb_%s.txt'%k) for k in ('r4base','new','nodyn3','trk1','trk2')}
for k in L['r4base']:
    a=L['r4base'][k]
    print('  %-16s r4 %6.2f | v6 %+5.1f %% | nodyn3 %+5.1f %% | maxima only %+5.1f %% | verdict + redo only %+5.1f %%'%(k,a,*[100*(L[v][k]/a-1) for v in ('new','nodyn3','trk1','trk2')]))
print("""  -> the four v_max3 per chunk cost +10 % on the long-K raw launches, the verdict and the redo pass nothing.  The ISA shows why: the refill load of the
     ring stage the maxima still read was hoisted above them into spare registers, followed by s_waitcnt vmcnt(0) + two v_mov into the stage -- the
     memory latency of every chunk exposed inside the pipelined loop.

r5c24 -- the final build (maxima pinned in front of a scheduling barrier, loads behind it):""")
a=_load('gpurun_out/r5c24_cb_r4.txt'); b=_load('gpurun_out/r5c24_cb_new.txt')
for k in a: print('  %-16s r4 %6.2f | final %6.2f (%+5.1f %%)'%(k,a[k],b[k],100*(b[k]/a[k]-1)))
