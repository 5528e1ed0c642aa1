#!/bin/bash
# Round 5, session 47: test_reference_replay with KNOWN_TIES pinned to one value
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_reference_replay.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4
