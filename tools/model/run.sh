#!/bin/bash
# bring-up: build and run the CPU model of the mask-resolve dfast parse against the oracle (seeds: first, count)
cd "$(dirname "$0")" && gcc -O2 -g -std=gnu11 -Wall -Wno-unused-function -I../../oracle -o /tmp/dfast_mask_model dfast_mask_model.c ../../oracle/zo_entropy.c ../../oracle/zo_decode.c -lm -ldl && /tmp/dfast_mask_model "$@"
