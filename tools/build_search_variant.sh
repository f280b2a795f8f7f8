#!/bin/bash
# build_search_variant.sh NAME [extra hipcc flags...]: tools/build_tu_variant.sh for the search translation units -- TU=search (one
# search), search_greedy (the device-resident greedy frontier of one search), search_many (many searches per call); default: search_greedy
exec "$(dirname "$0")/build_tu_variant.sh" "${TU:-search_greedy}" "$@"
