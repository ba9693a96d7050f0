#!/bin/bash
# phase timing of dds_layer_kernel at batch-1 size (tools/dds_micro.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
tools/bin/dds_micro "$@"
