#!/bin/bash
# GPU box: build tools/fetch_calib.hip, run it under the two counter passes, print counter bytes / known bytes per kernel
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/calib && /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/fetch_calib tools/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c --output-format csv -d /tmp/calib_$c -- /tmp/fetch_calib > /dev/null 2>&1; done
python3 - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/calib/fetch_calibration.txt
import csv, glob, collections
known = 1 << 30
print("rocprofv3 counter (KB x 1024) / known bytes, 1 GiB streamed per kernel (past the 256 MiB Infinity Cache), mean of 3 launches")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(dict)
    for f in glob.glob(f"/tmp/calib_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                d = acc[r["Kernel_Name"].split("(")[0]]
                d[r["Dispatch_Id"]] = d.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    for k, d in sorted(acc.items()):
        if "code_kernel" in k:  # per launch, in dispatch order: counter bytes / 128 KiB of code
            if c == "FETCH_SIZE":
                print(f"  {c:10s} {k:45s} per launch, counter bytes / 131072 code bytes: " + " ".join(f"{v * 1024.0 / 131072:.2f}" for _, v in sorted(d.items(), key=lambda kv: int(kv[0]))))
            continue
        v = sum(d.values()) / len(d) * 1024.0
        nb = known if "record" not in k else known // 192 * 192
        print(f"  {c:10s} {k:45s} {v / nb:6.3f}")
PY
