"""Parse a rocprofv3 --hip-runtime-trace --kernel-trace JSON: the HIP calls of one steady-state step in host order."""
import json, glob, sys
f = glob.glob(sys.argv[1] + "/tr/*/*_results.json")[0]
j = json.load(open(f))["rocprofiler-sdk-tool"][0]
bufs = j["buffer_records"]
ops = {}
for e in j["strings"]["buffer_records"]:
    ops[e["kind"] if isinstance(e["kind"], int) else e.get("kind")] = e["operations"]
kinds = {i: e for i, e in enumerate(j["strings"]["buffer_records"])}
ksym = {k["kernel_id"]: k.get("formatted_kernel_name") or k.get("demangled_kernel_name") or k.get("kernel_name") for k in j["kernel_symbols"]}
disp = {d["correlation_id"]["internal"]: d for d in bufs["kernel_dispatch"]}
api = sorted(bufs["hip_api"], key=lambda r: r["start_timestamp"])
def opname(r):
    return j["strings"]["buffer_records"][r["kind"]]["operations"][r["operation"]]
rows = []
for r in api:
    n = opname(r)
    d = disp.get(r["correlation_id"]["internal"])
    k = ksym.get(d["dispatch_info"]["kernel_id"], "?").replace("(anonymous namespace)::", "").split("(")[0][-50:] if d else ""
    rows.append((n, r.get("stream_id", {}).get("handle"), k, r))
idx = [i for i, x in enumerate(rows) if "ps_fwd_kernel" in x[2]] or [i for i, x in enumerate(rows) if "coattn_fwd_kernel" in x[2]]
pick = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[pick], idx[pick + 1]
# start the listing behind the previous step's last launch
while a > 0 and not rows[a - 1][2]:
    a -= 1
out = []
for n, s, k, r in rows[a:b]:
    if n in ("hipGetLastError", "hipPeekAtLastError", "hipGetDevice", "hipSetDevice", "hipGetDeviceCount", "hipDeviceGetAttribute",
             "hipEventQuery", "__hipPushCallConfiguration", "__hipPopCallConfiguration"):
        continue
    args = {a_["name"]: a_["value"] for a_ in r.get("args", [])}
    extra = ""
    if "Event" in n:
        extra = " ev=%s" % (args.get("event") or args.get("start") or "")[-6:]
    out.append("stream %-2s %-26s %s%s" % (s, n, k, extra))
open(sys.argv[1] + "/api_sequence.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))

# per step: how many records / waits the launch stream carries (the middle step above may be one that carries marks)
print()
for i in range(len(idx) - 1):
    seg = rows[idx[i]:idx[i + 1]]
    main = seg[0][1]
    nrec = sum(1 for n, s, k, r in seg if s == main and n == "hipEventRecord")
    nwait = sum(1 for n, s, k, r in seg if s == main and n == "hipStreamWaitEvent")
    ncreate = sum(1 for n, s, k, r in seg if n.startswith("hipEventCreate"))
    print("step %2d: launch-stream records %d waits %d, events created %d, launches %d" % (i, nrec, nwait, ncreate, sum(1 for x in seg if x[2])))
