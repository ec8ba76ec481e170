#!/bin/bash
# bring-up: build and run the CPU model of the round-4 link dfast parse against the oracle (args: first seed, count | file <path> <fs> <frames> <level>)
cd "$(dirname "$0")" && gcc -O2 -g -std=gnu11 -Wall -Wno-unused-function -I../../oracle -o /tmp/dfast_link_model dfast_link_model.c ../../oracle/zo_entropy.c ../../oracle/zo_decode.c -lm -ldl && /tmp/dfast_link_model "$@"
