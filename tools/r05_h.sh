#!/bin/bash
python __graft_entry__.py --smoke 2>&1 | tail -5
