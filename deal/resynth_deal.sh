#!/bin/bash
# MI355X counterpart of resyth_64sub_ori/deal.sh and resyth_64sub_IBM/deal.sh:
#   ./enhance_resyth_subband <cfg>       ratio (soft) mask
#   ./enhance_resyth_subband_IBM <cfg>   ideal binary mask
set -e
cd "$(dirname "$0")/.."
make -s -C speech_enhancement_amd/csrc
make -s -C speech_enhancement_amd/host
echo "built: speech_enhancement_amd/host/bin/enhance_resyth_subband{,_IBM}   (usage: <binary> <cfg>)"
