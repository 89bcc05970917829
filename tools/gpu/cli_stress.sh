#!/bin/bash
# the CLI with the device parsing tiny chunks, over and over: every run must end (no hang) with the golden output
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
CASE=${1:-c1_defaults}
D=tests/golden/$CASE
T=/tmp/stress_$$; mkdir -p $T
zcat $D/in.fq.gz > $T/in.fq
zcat $D/expected.out.fq.gz > $T/want.fq
FLAGS=$(python -c "import json; print(' '.join('$D/ADAPTERS.fa' if f == 'ADAPTERS.fa' else f for f in json.load(open('$D/case.json'))['flags']))")
bad=0; hung=0; n=0
for rep in $(seq 1 ${2:-40}); do
  for cb in 20000 30000 47000 100000 400000; do
    for rt in 1 3 8; do
      n=$((n+1))
      FPLH_CHUNK_BYTES=$cb timeout 30 bin/fastplong_amd -i $T/in.fq -o $T/out.fq -j $T/o.json -h $T/o.html --reader_threads $rt $FLAGS > $T/log 2>&1
      rc=$?
      if [ $rc -eq 124 ]; then hung=$((hung+1)); echo "HUNG cb=$cb rt=$rt"; fi
      if [ $rc -ne 0 ] || ! cmp -s $T/out.fq $T/want.fq; then bad=$((bad+1)); echo "BAD rc=$rc cb=$cb rt=$rt"; tail -3 $T/log; fi
    done
  done
done
echo "$CASE: $n runs, $bad bad, $hung hung"
rm -rf $T
