"""ORACLE — TEST INFRASTRUCTURE ONLY. See oracle/oracle.py."""
