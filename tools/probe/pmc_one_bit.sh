# HBM traffic of the one-bit product kernel (library defaults on the 27-pt 256^3 block), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
# separate passes (gfx950: FETCH_SIZE in 32-byte... see MI355X_MICROARCH.md: KiB units, doubled), per launch.  bash tools/probe/pmc_one_bit.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc1_$c
  PA_K1_VD=1 timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc1_$c -o c -- python $R/tools/probe/k1_time.py 256 2 > /tmp/pmc1_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc1_{c}/**/*counter_collection.csv", recursive=True)
    acc, n = 0.0, 0
    for r in csv.DictReader(open(f[0])):
        if "k_spmv_pell" in r["Kernel_Name"] and r["Counter_Name"] == c:
            acc += float(r["Counter_Value"]); n += 1
    res[c] = (acc / max(n, 1), n)
fetch = res["FETCH_SIZE"][0] * 1024 * 2      # KiB units, doubled on gfx950 (MI355X_MICROARCH.md)
write = res["WRITE_SIZE"][0] * 1024
print(f"one bit per entry, 27-pt 256^3: per launch FETCH {fetch/1e6:.1f} MB + WRITE {write/1e6:.1f} MB = {(fetch+write)/1e6:.1f} MB over {res['FETCH_SIZE'][1]} launches")
PY
