#!/bin/bash
# Kill stray training ranks on every node after a hang (intent of the reference's
# process_cleanup.sh; that script's `ssh -p $node && pkill -9 python` is malformed and would hit
# every python process of the user).  The native launcher registers the exact PID of each rank it
# starts under /tmp/dpa_ranks_<uid>/ and already tears all ranks down when one fails, so this is
# a last resort that signals only those registered PIDs.
#   usage: NODES="hal01 hal02" scripts/process_cleanup.sh          (default: this host only)
CLEAN='d=${DPA_RUN_DIR:-/tmp}/dpa_ranks_$(id -u); for f in "$d"/*; do [ -e "$f" ] || continue; pid=$(basename "$f"); if [ -d /proc/$pid ]; then kill -9 "$pid"; fi; rm -f "$f"; done'
NODES=${NODES:-localhost}
for node in $NODES; do
  if [ "$node" = "localhost" ]; then
    bash -c "$CLEAN"
  else
    ssh "$node" "$CLEAN" &
  fi
done
wait
