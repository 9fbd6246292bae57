#!/bin/bash
# MI355X counterpart of the reference's etsi/deal.sh: builds the engine and the file-in/file-out
# driver, which is then run exactly like the reference binary:   ./etsi_denoise <cfg>
# (cfg format: etsi/cpp/main.cpp:73-141)
set -e
cd "$(dirname "$0")/.."
make -s -C speech_enhancement_amd/csrc
make -s -C speech_enhancement_amd/host
echo "built: speech_enhancement_amd/host/bin/etsi_denoise   (usage: etsi_denoise <cfg>)"
