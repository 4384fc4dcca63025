import csv, glob, collections, sys
rows = collections.OrderedDict()
for p in sys.argv[1:]:
    fs = sorted(glob.glob(f'gpurun_out/pmc_{p}/*/*_counter_collection.csv'))
    f = fs[-1]
    for r in csv.DictReader(open(f)):
        d = int(r['Dispatch_Id'])
        e = rows.setdefault(d, {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']), 'lds': int(r['LDS_Block_Size'])})
        e[r['Counter_Name']] = float(r['Counter_Value'])
        e.setdefault('dur_' + p, float(r['End_Timestamp']) - float(r['Start_Timestamp']))
agg = collections.OrderedDict()
for d, e in rows.items():
    if 'k_conv_mfma' not in e['name']: continue
    k = (e['name'].split('(')[0][-44:], e['grid'] // 256, e['lds'])
    a = agg.setdefault(k, collections.Counter()); a['n'] += 1
    for c, v in e.items():
        if isinstance(v, float): a[c] += v
def g(a, c): return a.get(c, 0.0) / a['n']
print(f"{'kernel':44s} {'wgs':>5s} {'lds':>6s}  n  dur_us | L1acc/clk/CU  L1->L2 req/clk/CU  TCCreq/clk/CU  avgL2lat  TApend%  TA_busy% | vmem_level  vmem_rd/wave  addrfifo_full% cmdfifo_full% | wait_any%")
for k, a in sorted(agg.items(), key=lambda kv: -g(kv[1], 'dur_A') * kv[1]['n'])[:16]:
    dur = g(a, 'dur_A') / 1e3; clk = dur * 1e-6 * 2.4e9
    l1 = g(a, 'TCP_TOTAL_CACHE_ACCESSES'); req = g(a, 'TCP_TCC_READ_REQ'); lat = g(a, 'TCP_TCC_READ_REQ_LATENCY')
    waves = g(a, 'SQ_WAVES') or 1; wc = g(a, 'SQ_WAVE_CYCLES') or 1
    print(f"{k[0]:44s} {k[1]:5d} {k[2]:6d} {a['n']:2d} {dur:7.1f} | {l1/clk/256:8.3f} {req/clk/256:14.3f} {g(a,'TCC_REQ')/clk/256:14.3f} {lat/max(1,req):9.0f} {100*g(a,'TCP_PENDING_STALL_CYCLES')/clk/256:7.1f} {100*g(a,'GRBM_TA_BUSY')/max(1,g(a,'GRBM_GUI_ACTIVE')):7.1f} | "
          f"{g(a,'SQ_INST_LEVEL_VMEM')/wc:8.2f} {g(a,'SQ_INSTS_VMEM_RD')/waves:10.0f} {100*g(a,'SQ_VMEM_TA_ADDR_FIFO_FULL')/wc:10.1f} {100*g(a,'SQ_VMEM_TA_CMD_FIFO_FULL')/wc:10.1f} | {100*g(a,'SQ_WAIT_ANY')/wc:6.1f}")
