# A/B of library builds in ONE box session: tools/ab_chains.sh <name>=<libocrf_hip.so> ...   (in-tree build first)
# Every round runs tools/chains_r5.py once per build; three rounds.
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  echo "== round $rep, in-tree"; timeout -k 10 150 python tools/chains_r5.py 2>&1 | grep -E "^step|grid  *(0|704):"
  for kv in "$@"; do
    echo "== round $rep, ${kv%%=*}"; OCRF_HIP_SO=${kv#*=} timeout -k 10 150 python tools/chains_r5.py 2>&1 | grep -E "^step|grid  *(0|704):"
  done
done
