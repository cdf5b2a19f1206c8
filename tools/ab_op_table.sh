#!/bin/bash
# same-box A/B of library variants: tools/ab_op_table.sh <outdir> <name> [<name> ...]   (name "default" = the in-tree library)
# prints the per-kind sums of tools/op_table.py for each, twice, interleaved
OUT=$1; shift
mkdir -p $OUT
for rep in 1 2; do
  for n in "$@"; do
    if [ "$n" = default ]; then unset SR3_LIBRARY; else export SR3_LIBRARY=$PWD/tools/bin/libsr3_$n.so; fi
    python tools/op_table.py $AB_OPTS > $OUT/op_table_${n}_$rep.txt 2> $OUT/op_table_${n}_$rep.err
    echo "== $n (rep $rep): $(grep -E '^#  (555|565|575|653) ' $OUT/op_table_${n}_$rep.txt | tr -s ' ' | tr '\n' ';')"
  done
done
