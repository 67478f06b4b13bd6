// placeholder, filled in next
