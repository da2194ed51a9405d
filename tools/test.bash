#!/bin/bash
# The reference's batch driver, generate_signatures/test.bash:4-22 + launch/lidar.launch:13-21, for the MI355X executables:
# every KITTI / RobotCar sequence x every lidar descriptor (delight, m2dp, sc), each run with the parameters lidar.launch sets
# (poses_history_file, pts_history_file, <method>_file, incoming_id_file, lidarRange = 45.0) under
# <results>/<dataset>/<seq>/.  The reference's gist.launch / bow.launch lines (image descriptors) have no counterpart here:
# SURVEY.md §8 scopes the generators to the lidar path; their matchers are in match_signatures --type gist|bow.
#
# usage: tools/test.bash [results_dir] [extra test_<method> arguments, e.g. _device:=1 or _gpu_prestage:=0]
#   results_dir defaults to ./results (the layout of place_recognition/results in the reference)
#   PR_BIN overrides the directory of the executables (default: so_dso_place_recognition_amd/bin next to this script)
set -u
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
BIN="${PR_BIN:-$HERE/../so_dso_place_recognition_amd/bin}"
RESULTS="${1:-results}"
shift || true
rc=0
run() {   # dataset seq method
  local folder="$RESULTS/$1/$2"
  if [ ! -f "$folder/poses_history_file.txt" ] || [ ! -f "$folder/pts_history_file.txt" ]; then
    echo "skip $1/$2 ($3): poses_history_file.txt / pts_history_file.txt missing" >&2
    return
  fi
  "$BIN/test_$3" "_poses_history_file:=$folder/poses_history_file.txt" "_pts_history_file:=$folder/pts_history_file.txt" \
                 "_$3_file:=$folder/history_$3.txt" "_incoming_id_file:=$folder/incoming_id_file.txt" "_lidarRange:=45.0" "${EXTRA[@]}" || rc=1
}
EXTRA=("$@")
# KITTI (test.bash:4-12)
for s in seq00 seq02 seq05 seq06 seq07; do
  for m in delight m2dp sc; do run KITTI "$s" "$m"; done
done
# RobotCar (test.bash:14-22)
for s in 2014-07-14-14-49-50 2014-11-28-12-07-13 2014-12-12-10-45-15 2015-02-10-11-58-05 2015-05-19-14-06-38 2015-05-22-11-14-30 2015-08-13-16-02-58 2015-10-30-13-52-14; do
  for m in delight sc m2dp; do run RobotCar "$s" "$m"; done
done
exit $rc
