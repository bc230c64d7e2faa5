#!/bin/bash
# One GPU lease = one recipe.  usage: tools/lease.sh <recipe> [timeout seconds] [recipe args...]
#   runs tools/recipes/<recipe>.sh on a fresh MI355X box through gpurun (from this container), or directly when already on the box
#   (GRAFT_REPO_ROOT set).  Recipes write under gpurun_out/; what is to be judged is copied to profiles/ by hand.
recipe=$1; shift
timeout=${1:-1500}; shift
here=$(cd "$(dirname "$0")" && pwd)
[ -f "$here/recipes/$recipe.sh" ] || { echo "no such recipe: $recipe (have: $(ls $here/recipes | sed 's/\.sh$//' | tr '\n' ' '))"; exit 2; }
if [ -n "$GRAFT_REPO_ROOT" ]; then exec bash "$here/recipes/$recipe.sh" "$@"; fi
args=$(printf '%q ' "$@")
exec /usr/local/graft/bin/gpurun --timeout "$timeout" -- "bash tools/recipes/$recipe.sh $args"
