import json, collections, sys
def load(l):
    ops = json.load(open(f'gpurun_out/ops_{l}.json'))
    agg = collections.OrderedDict()
    for o in ops:
        d = o['desc'].split(' tile')[0]
        k = (o['kind'], d)
        a = agg.setdefault(k, [0, 0.0, 0.0, set(), 0.0])
        a[0] += 1; a[1] += o['ms']; a[2] += o['gflop']; a[4] += o['mbytes']
        a[3].add(o['desc'].split(' tile')[-1] if ' tile' in o['desc'] else '')
    return agg
labels = sys.argv[1:]
aggs = [load(l) for l in labels]
rows = sorted(aggs[0].items(), key=lambda kv: -kv[1][1])[:int(__import__('os').environ.get('TOP', '45'))]
print(f"{'op':42s} n " + " ".join(f"{l[:14]:>14s}" for l in labels))
for k, v in rows:
    cells = []
    for a in aggs:
        if k in a:
            x = a[k]; cells.append(f"{x[1]:6.3f} {x[2]/x[1] if x[1] else 0:4.0f}TF")
        else: cells.append(" " * 13)
    print(f"{k[1]:42s} {v[0]:2d} " + "  ".join(cells) + "   " + ",".join(sorted(v[3])) + f"  {v[4]/v[1]:.0f}GB/s")
for l, a in zip(labels, aggs):
    bk = collections.Counter()
    for k, v in a.items(): bk[k[0]] += v[1]
    print(l, {k: round(v, 3) for k, v in bk.items()}, "total", round(sum(bk.values()), 3))
