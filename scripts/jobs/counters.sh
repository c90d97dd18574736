rocprofv3 -L 2>/dev/null | grep -E "Counter_Name" | sed 's/.*:\s*//' | sort -u | tr '\n' ' ' | fold -w 200
