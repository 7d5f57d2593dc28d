"""Replay of the rolling lock-step driver (two lanes of 19 slots, shared initial scores, tail hand-over) on the iteration counts a
real run recorded (MENDELIHT_CV_TRACE=1 tools/cv_trace.sh -> profiles/r05_cv_trace.txt): how many fused passes does a queue order
cost?  A pass is 17.7 ms + 1.07 ms per residual, the residual scores are fixed, so passes are what an order can save."""
import re, collections, itertools, random
rows=[]
import os
for ln in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r05_cv_trace.txt')):
    m=re.match(r"fit out_index (\d+) k (\d+): (\d+) iterations", ln)
    if m: rows.append(tuple(map(int,m.groups())))
first=rows[:100]
it={oi:itn for oi,k,itn in first}      # out_index = fold*20 + (k-1)
def cost(m): return 17.7+1.07*m
def simulate(order, cap=19, lanes=2, merge=True, verbose=False):
    queue=list(order); qi=0
    slots=[[None]*cap for _ in range(lanes)]     # each: [fit, remaining_scores, fresh]
    cached=[set() for _ in range(lanes)]
    passes=0; resid=0; t=0.0; hist=[]
    active_l=[True]*lanes
    drained=False
    while True:
        progressed=False
        for L in range(lanes):
            if not active_l[L]: continue
            # handover
            if merge and L==1 and drained:
                mine=[s for s in slots[1] if s]
                free0=sum(1 for s in slots[0] if s is None)
                if mine and len(mine)<=free0:
                    for s in mine:
                        i=slots[0].index(None); slots[0][i]=s
                    slots[1]=[None]*cap; active_l[1]=False
                    continue
            need=0
            for i in range(cap):
                while True:
                    s=slots[L][i]
                    if s is None:
                        if qi>=len(queue): drained=True; break
                        f=queue[qi]; qi+=1
                        fold=f//20
                        if fold in cached[L]:
                            slots[L][i]=[f,it[f]]   # init served by copy; first step rides this round
                        else:
                            slots[L][i]=[f,it[f]+1]; cached[L].add(fold)
                        continue
                    if s[1]==0: slots[L][i]=None; continue
                    break
                if slots[L][i] is not None: need+=1
            if need==0:
                active_l[L]=False; continue
            passes+=1; resid+=need; t+=cost(need); hist.append(need); progressed=True
            for i in range(cap):
                if slots[L][i] is not None: slots[L][i][1]-=1
        if not progressed: break
    return passes, resid, t, hist
fold_major = list(range(100))
byk = {f % 20: it[f] for f in range(20)}
orders = {"fold-major (the caller's order)": fold_major,
          "clairvoyant longest first": sorted(range(100), key=lambda f: -it[f]),
          "by the first fold's counts": sorted(range(100), key=lambda f: (-byk[f % 20], f)),
          "k descending, folds interleaved": sorted(range(100), key=lambda f: (-(f % 20), f // 20)),
          "first 38 fold-major, the rest by known counts": fold_major[:38] + sorted(fold_major[38:], key=lambda f: (-byk[f % 20], f)),
          "shortest first": sorted(range(100), key=lambda f: it[f])}
full = dict(it)
for label, counts in (("every step scored (rounds 1-4)", {f: full[f] for f in full}),
                      ("the converging step's score skipped (round 5)", {f: full[f] - 1 for f in full})):
    it = counts
    print(label)
    for name, order in orders.items():
        p_, r_, t_, h_ = simulate(order)
        print(f"  {name:48s} {p_} passes, {r_} residuals, {t_:7.0f} ms of passes, {r_ / p_:5.2f} residuals per pass")
    print(f"  (if every pass were full: {-(-(sum(counts.values()) + 10) // 19)} passes)")
