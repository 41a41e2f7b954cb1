"""Import stub (test infrastructure only): lets the unmodified reference under /root/reference import
in a container without gym3. Only the names the reference's hot path touches are provided."""
