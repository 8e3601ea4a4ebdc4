#!/bin/bash
# the round's PMC collections: one pmc_passes.sh (four separate rocprofv3 --pmc runs) per kernel family
set -o pipefail
bash pmc_passes.sh iau2 conv2 'imgActs+maxUndo' &&
bash pmc_passes.sh iau3 local3 'imgActs+maxUndo' &&
bash pmc_passes.sh fa conv filterActs &&
bash pmc_passes.sh fc6 fc6 '' &&
bash pmc_passes.sh wa1 conv1 weightActs &&
bash pmc_passes.sh wa2 conv2 weightActs &&
bash pmc_passes.sh ia conv 'imgActs ' 
