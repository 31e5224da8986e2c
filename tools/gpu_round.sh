# usage: bash tools/gpu_round.sh <tag> [nopmc]  -- the evidence of a round in ONE call, everything into gpurun_out/<tag>_*:
#   PMC passes of the Jacobian / assembly / factorisation kernels, stamped with the build head (tools/pmc_round.sh) -- FIRST, so that
#   the bench line below can cite them (copy them to profiles/ before the final bench if the line is to name committed files);
#   full bench line; rocprofv3 kernel summary of the same command; kernel timeline of one LM iteration; C4 line + its kernel summary;
#   rocSOLVER cross-check; factorisation size sweep; the dense-visibility table and bench line.
# Regenerating profiles/ for a round = this script + `cp gpurun_out/<tag>_* profiles/`.
export TMPDIR=/tmp
TAG=${1:-tmp}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
if [ "$2" != "nopmc" ]; then
  bash $R/tools/pmc_round.sh $TAG 6000 24000 > $O/${TAG}_pmc.log 2>&1
  # the bench line cites the newest profiles/r<N>_pmc_*.json: on the box, put this run's files there
  for k in jacobian assembly_kernels chol_mfma; do [ -f $O/${TAG}_pmc_$k.json ] && cp $O/${TAG}_pmc_$k.json $R/profiles/${TAG}_pmc_$k.json; done
fi
python $R/bench.py > $O/${TAG}_bench_full.json 2> $O/${TAG}_bench_full.err
P=/tmp/prof_$TAG; rm -rf $P; mkdir -p $P
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python $R/bench.py --steps 20 --reps 2 --warmup 2 --no-cpu-baseline --no-library-baseline --no-drop-in > $P/bench.json 2> $P/bench.err)
find $P -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_kernel_stats.csv
bash $R/tools/gpu_iter_trace.sh > $O/${TAG}_iter_trace.txt 2>&1
bash $R/tools/gpu_c4.sh $TAG > $O/${TAG}_c4.log 2>&1
python $R/tools/rocsolver_potrf.py 6000 12000 24000 > $O/${TAG}_rocsolver_potrf.txt 2>&1
python $R/tools/size_sweep.py 3000 6000 9000 12000 16000 24000 > $O/${TAG}_size_sweep.txt 2>&1
# dense visibility: the two forms of the Schur complement side by side, and the landmark-heavy bench line (not a BASELINE config)
python $R/tools/dense_schur_time.py 29 600 60 12000 100 8000 300 8000 1000 20000 100 200000 2>&1 | grep -v amdgpu.ids > $O/${TAG}_dense_schur.txt
python $R/bench.py --cams 100 --pts 200000 --dense-visibility --no-cpu-baseline --no-library-baseline --steps 20 --reps 3 > $O/${TAG}_dense_bench.json 2> $O/${TAG}_dense_bench.err
head -12 $O/${TAG}_kernel_stats.csv | cut -c1-160
tail -3 $O/${TAG}_bench_full.err
cat $O/${TAG}_size_sweep.txt
python - <<PY
import json
d=json.loads(open("$O/${TAG}_bench_full.json").read().strip().splitlines()[-1])
print('it/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'phases', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['phase_ms_per_step'].items()})
print('roofline frac', d['roofline']['frac'], 'jac frac', d['roofline_jacobian']['frac'], 'solo', d['roofline_jacobian']['solo']['frac'], 'cpu', d['cpu_baseline']['value'], 'speedup', d.get('speedup_vs_cpu_port'), 'gate', d['matched_result_gate']['passed'])
print('library', d.get('library_baseline')); print('traffic sources', d['roofline']['traffic_source']['same_sources_as_this_build'], d['roofline_jacobian']['traffic_source']['same_sources_as_this_build'], d['roofline_schur']['traffic_source']['same_sources_as_this_build'], d['roofline_schur']['traffic'])
print('predicted', d.get('predicted_scaling', {}).get('speedup_at_8_best'), d.get('predicted_scaling', {}).get('speedup_at_8_ring'))
PY
tail -4 $O/${TAG}_c4.log
