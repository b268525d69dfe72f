cd /root/repo; timeout 1500 python -m pytest tests/ -q -m gpu --durations=6 -x --deselect tests/test_bench_contract.py::test_bench_json_line_contract 2>&1 | tail -30
