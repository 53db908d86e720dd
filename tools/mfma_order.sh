#!/bin/bash
# Does the operand sequence of back-to-back MFMAs change the power-limited rate?  (tools/mfma_peak order 0..3, interleaved)
for rep in 1 2 3; do
  for data in 0 2; do
    for order in 0 1 2 3; do
      tools/ubench/mfma_peak 2 20000 $data $order | tail -n 1
    done
  done
done
