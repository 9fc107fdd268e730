#!/bin/bash
# Same-box A/B of library builds with an arbitrary command:  bash tools/ab_cmd.sh "<command>" <lib1.so> <lib2.so> ...
CMD=$1; shift
LIVE=event_3dgs_amd/libe3dgs_hip.so
cp $LIVE /tmp/live.so
for src in "$@"; do
  cp $src $LIVE
  echo "== $src"; bash -c "$CMD" 2>&1 | tail -2
done
cp /tmp/live.so $LIVE
