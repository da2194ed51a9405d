bash tools/abn.sh 2 dps3 dps6 dps12 2>&1 | sed 's/select.*//'
