#!/bin/bash
exec "$(dirname "$0")/../build_tu_variant.sh" shard "$@"
