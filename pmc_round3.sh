#!/bin/bash
# round 3 PMC collections (same four passes per family as pmc_round.sh); $1 = a | b
set -o pipefail
if [ "$1" = "a" ]; then
bash pmc_passes.sh iau2 conv2 'imgActs+maxUndo' &&
bash pmc_passes.sh iau3 local3 'imgActs+maxUndo' &&
bash pmc_passes.sh fa local filterActs &&
bash pmc_passes.sh fc6 fc6 '' &&
bash pmc_passes.sh fc5 fc5 '' &&
bash pmc_passes.sh wa1 conv1 weightActs &&
bash pmc_passes.sh wa2 conv2 weightActs
elif [ "$1" = "c" ]; then
bash pmc_passes.sh wa1 'conv1$' weightActs &&
bash pmc_passes.sh wa2 'conv2$' weightActs &&
bash pmc_passes.sh wa2p 'conv2+' weightActs &&
bash pmc_passes.sh fc6 'fc6$' ''
else
bash pmc_passes.sh wa3 local3 weightActs &&
bash pmc_passes.sh wa4 local4 weightActs &&
bash pmc_passes.sh ia4 local4 'imgActs' &&
MCN_IA_WIDE=0 bash pmc_passes.sh ian 'local3+' 'imgActs' &&
MCN_IA_WIDE=0 bash pmc_passes.sh ian2 'conv2+' 'imgActs'
fi
