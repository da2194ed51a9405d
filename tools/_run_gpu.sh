for v in bd6 ad4 ld1 vs1 vs2 vs3; do echo "== $v"; bash tools/ab.sh $v 2; done
