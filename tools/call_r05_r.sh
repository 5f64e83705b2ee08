set -x
# Montgomery-form stores: every immediate below 2^16 through one batched table load (default) against round 2-4's per-half branches
bash tools/gpu_call.sh r05_r "tests:mont or handoff or stage" benchq "benchq:--montgomery 1" env:ZKWG_MONT_BRANCHY=1 "benchq:--montgomery 1" env:ZKWG_MONT_BRANCHY=0 abc
mv gpurun_out/r05_r_abc.json gpurun_out/r05_r_abc_tables.json
bash tools/gpu_call.sh r05_r env:ZKWG_MONT_BRANCHY=1 abc
mv gpurun_out/r05_r_abc.json gpurun_out/r05_r_abc_branchy.json
