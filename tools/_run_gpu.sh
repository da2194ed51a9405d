for v in dp80 dp40x4 dp100; do echo "== $v"; bash tools/ab.sh $v 2; done
