#!/bin/bash
# build_search_variant.sh NAME [extra hipcc flags...]: tools/build_tu_variant.sh for acx_search.hip
exec "$(dirname "$0")/build_tu_variant.sh" search "$@"
