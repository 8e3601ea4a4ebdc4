#!/bin/bash
timeout -k 10 300 python -u tools/flip_probe.py
