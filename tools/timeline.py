"""Timeline of the last step of a rocprofv3 kernel trace: union busy time, idle gaps > 20 us, phase markers."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pre = [i for i, r in enumerate(rows) if "preprocess_kernel" in r["Kernel_Name"]]
step = rows[pre[-4]:]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]) for r in step)
t0 = iv[0][0]; cur_e = iv[0][1]; busy = 0; cur_s = iv[0][0]; gaps = []
last_name = iv[0][2]
for s, e, n in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        if s - cur_e > 20000: gaps.append(((cur_e - t0) / 1e6, (s - cur_e) / 1e3, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n
busy += cur_e - cur_s
print("union busy ms %.3f  span ms %.3f" % (busy / 1e6, (cur_e - t0) / 1e6))
for g in gaps[:40]: print("gap at %.3f ms: %.1f us  after %s  before %s" % g)
marks = ("radix_sort", "nms_scan", "roi_align_fwd", "wsddn", "roi_align_bwd_gather", "sgd_kernel", "multi_prep", "maxpool", "rpn_loss")
for s, e, n in iv:
    if any(m in n for m in marks): print("%8.3f ms (+%.0f us) %s" % ((s - t0) / 1e6, (e - s) / 1e3, n))
