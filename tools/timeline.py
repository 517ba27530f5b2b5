import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))) for r in rows)
# find step boundaries: multi_tensor_apply (Adam) kernels mark step ends
adam = [i for i, e in enumerate(ev) if "FusedOptimizer" in e[2] or "multi_tensor_apply" in e[2] or "adam_kernel" in e[2] or "sgd_kernel" in e[2]]
# group consecutive adam kernels
ends = []
for i in adam:
    if not ends or i - ends[-1][-1] > 5: ends.append([i])
    else: ends[-1].append(i)
print("steps seen:", len(ends))
a, b = ends[-3][-1] + 1, ends[-2][-1] + 1
step = ev[a:b]
t0, t1 = step[0][0], max(e[1] for e in step)
print("step wall %.3f ms, kernels %d" % ((t1 - t0) / 1e6, len(step)))
# union busy
busy = 0; cur_s, cur_e = None, None
for s, e, _, _ in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in step)
print("busy (union) %.3f ms, sum of kernel durations %.3f ms, idle %.3f ms" % (busy / 1e6, tot / 1e6, (t1 - t0 - busy) / 1e6))
# gaps histogram
gaps = []
cur_e = step[0][1]
for s, e, nme, _ in step[1:]:
    if s > cur_e: gaps.append((s - cur_e, nme))
    cur_e = max(cur_e, e)
gaps.sort(reverse=True)
print("largest gaps (us):", [(round(g / 1e3, 1), n[:50]) for g, n in gaps[:8]])
print("gaps total %.3f ms over %d gaps" % (sum(g for g, _ in gaps) / 1e6, len(gaps)))
by = collections.defaultdict(float)
for s, e, n, q in step: by[q] += (e - s)
print("per stream/queue busy ms:", {k: round(v / 1e6, 3) for k, v in by.items()})
# per-stream, per-kernel-family totals of the analysed step
fam = collections.defaultdict(float)
for s, e, n, q in step:
    key = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    fam[(q, key)] += (e - s) / 1e6
for (q, key), v in sorted(fam.items(), key=lambda kv: -kv[1])[:40]:
    print("stream %s  %-62s %.3f ms" % (q, key, v))
# tail of the backward pass: after the compute stream's last kernel, how long until the optimizer can start (side stream catching up)
opt_i = [i for i, e in enumerate(step) if "adam_kernel" in e[2] or "sgd_kernel" in e[2] or "FusedOptimizer" in e[2]]
if opt_i:
    o = opt_i[0]
    main_q = step[o][3]
    prev_main = [e for e in step[:o] if e[3] == main_q]
    prev_side = [e for e in step[:o] if e[3] != main_q]
    if prev_main and prev_side:
        t_main = max(e[1] for e in prev_main); t_side = max(e[1] for e in prev_side)
        print("backward tail: compute stream done %.3f ms before the optimizer starts; side stream done %.3f ms before; side lags the compute stream by %.3f ms"
              % ((step[o][0] - t_main) / 1e6, (step[o][0] - t_side) / 1e6, (t_side - t_main) / 1e6))
        # last few side kernels
        tail = sorted(prev_side, key=lambda e: e[1])[-6:]
        for s_, e_, n_, q_ in tail:
            print("   side tail kernel %-60s start %+8.1f us end %+8.1f us (relative to the compute stream's last kernel end)"
                  % (n_.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60], (s_ - t_main) / 1e3, (e_ - t_main) / 1e3))
