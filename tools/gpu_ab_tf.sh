# A/B of the temporal-filter motion search at a given bit depth: bash tools/gpu_ab_tf.sh "<lib> <lib> ..." <bit depth>
for round in 1 2; do
for L in $1; do
  echo "== $L bd=$2 round $round"
  AOMHIP_LIB=$L python - $2 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import bench, aom_av1_psy_amd as pkg, pyoracle as orc
ctx = pkg.capi.Context(0)
r = bench.run_tf(pkg, ctx, orc if len(sys.argv) > 2 else None, 8, 1, bd=int(sys.argv[1]))
print({k: r[k] for k in ("q30_mesh_pruned_when_close", "q12_mesh_always", "parity_sample")})
PY
done
done
