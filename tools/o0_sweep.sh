# A/B of the descriptor-driven streaming kernel's variants on one box: complete --O0 witnesses (tools/bench_full.py) and the
# prover stage from the image (tools/bench_abc.py) under the environment settings listed in $@ (one quoted string each)
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/bench_full.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); o=d['complete O0']; print('O0', o['witnesses_per_s'], o['GBps_written'], o['kernel_ms'])"
  env $cfg timeout 300 python tools/bench_abc.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())['from_image']; print('abc std', d['standard']['witnesses_per_s'], d['standard']['kernel_ms'], 'mont', d['montgomery']['witnesses_per_s'], d['montgomery']['kernel_ms']['zk_expand'])"
done
