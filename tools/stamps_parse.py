"""Pretty-print tools/stamps_batch.sh logs: per layer, compute-wave and loader-wave phase durations in cycles."""
import re, sys
txt = open(sys.argv[1]).read()
for sec in txt.split('=== op')[1:]:
    lines = sec.strip().splitlines()
    print('OP', lines[0])
    for l in lines[1:]:
        if l.startswith('op:'): print(l)
    st = {}
    for l in lines:
        m = re.match(r'stamp\s+(\d+): \+\s*(\d+)', l)
        if m: st[int(m.group(1))] = int(m.group(2))
    if 64 in st:
        prev = st.get(0, 0); out = []
        for v in range(20):
            a, b, e = st.get(1 + 3 * v), st.get(2 + 3 * v), st.get(3 + 3 * v)
            if a is None: break
            s = f"v{v}: mfma {a - prev} bar {b - a}"
            prev = b
            if e is not None and e > b: s += f" epi {e - b}"; prev = e
            out.append(s)
        print(' C start', st.get(0), '|', ' ; '.join(out))
        prev = st[64]; out = []
        for v in range(15):
            k = [st.get(65 + 4 * v + i) for i in range(4)]
            if None in k: break
            out.append(f"v{v}: issue {k[0] - prev} conv {k[1] - k[0]} fetch {k[2] - k[1]} bar {k[3] - k[2]}")
            prev = k[3]
        print(' L start', st[64], '|', ' ; '.join(out))
    else:
        ks = sorted(st.items()); prev = 0
        print(' ', ' '.join(f"[{i}] +{v - prev}" for (i, v), prev in zip(ks, [0] + [v for _, v in ks[:-1]])))
