./tools/probe/valu_rate
