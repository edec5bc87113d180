"""CPU oracle (test infrastructure).  Importable only from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); the product package never imports it."""
