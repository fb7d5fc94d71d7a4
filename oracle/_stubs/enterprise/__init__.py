"""Stand-in package for enterprise — TEST INFRASTRUCTURE ONLY."""
