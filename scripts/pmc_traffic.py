"""HBM traffic of the implicit-GEMM launches of the headline bench step from two rocprofv3 --pmc passes.

  cd /tmp && rocprofv3 --pmc FETCH_SIZE -d <dir>/fetch -o out --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras
  cd /tmp && rocprofv3 --pmc WRITE_SIZE -d <dir>/write -o out --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras
  python scripts/pmc_traffic.py <dir>/fetch/out_counter_collection.csv <dir>/write/out_counter_collection.csv > profiles/<name>.json

The bench runs one set-up pass, one warm-up step and one timed step: the set-up pass launches the same kernels,
so three passes of the step are in the trace (third argument, default 3).  With the two-stream batch split a step has about
twice the launches of a one-stream step (each on half a batch): the record therefore also gives the bytes PER STEP, which do
not depend on how the step is cut into launches."""
import csv, json, sys


def total(path, counter):
    s, n = 0.0, 0
    for row in csv.DictReader(open(path)):
        if "igemm_kernel" in row["Kernel_Name"] and row["Counter_Name"] == counter:
            s += float(row["Counter_Value"])
            n += 1
    return s, n


fetch_kb, nf = total(sys.argv[1], "FETCH_SIZE")
write_kb, nw = total(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
read_b = fetch_kb * 1024 * 2 / nf        # gfx950: 16-byte-per-lane reads are tallied at half their size
write_b = write_kb * 1024 / nw
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3     # set-up pass + warm-up step + timed step
launches_per_step = nf // passes
print(json.dumps({
    "what": "HBM traffic of the igemm_kernel launches of one headline bench step (1024^2, batch 16, 1 NFE), rocprofv3 --pmc, separate passes",
    "command": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras   (second pass: --pmc WRITE_SIZE)",
    "workload": {"res": 1024, "batch": 16, "nfe": 1},
    "igemm_launches": nf, "FETCH_SIZE_KB_sum": fetch_kb, "WRITE_SIZE_KB_sum": write_kb,
    "correction": "gfx950: FETCH_SIZE tallies the 16-byte-per-lane buffer_load...lds reads at half their size -> x2; WRITE_SIZE exact",
    "read_bytes_per_launch": read_b, "write_bytes_per_launch": write_b, "traffic_bytes_per_launch": read_b + write_b,
    "passes_in_trace": passes, "igemm_launches_per_step": launches_per_step,
    "traffic_bytes_per_step": (read_b + write_b) * launches_per_step,
    "per_step_GB": {"read": round(read_b * launches_per_step / 1e9, 1), "write": round(write_b * launches_per_step / 1e9, 1)},
}, indent=1))
